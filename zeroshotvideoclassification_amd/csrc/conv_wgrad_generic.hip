// conv_wgrad_generic.hip -- weight gradient, (channel, tap)-ordered columns, for layers whose
// input has fewer than 16 channels (the 3-channel stems: resnet.py:170,181; network.py:102).
//
// Same GEMM, MFMA, slab reduction and LDS images as conv_wgrad.hip, but a column is one
// (ci, tap) pair in the weight tensor's own order, so every column carries its own tap: the
// per-voxel padding mask is tested per gathered element.  With 3 input channels the tap-major
// kernel would pad every tap's channel block from 3 to 16 (5x wasted MFMAs); here K = 147 is
// padded to 192 at most.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_params.h"
#include "knobs.h"

namespace zsv {

struct WgradGenParams {
    int M;            // Cout
    int K;            // Cin * taps
    int P;            // N * oS voxels
    int taps, kHW, kW, kH, kT;
    int oS, oHW, oW;  // dY geometry
    int gC, gT, gH, gW, gS, gHW;   // x geometry
    int sT, sH, sW, pT, pH, pW;
    int chunks_per_slice;          // 32-voxel chunks handled by one slice
    unsigned x_bytes, dy_bytes;    // buffer sizes for the hardware range check
};

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_wgrad_generic_kernel(WgradGenParams prm, const float* __restrict__ X,
                                                         const float* __restrict__ DY,
                                                         float* __restrict__ OUT, int tiles_m) {
    constexpr int BM = 16 * TM * WGM;
    constexpr int BN = 16 * TN * WGN;
    constexpr int BP = 32;                 // voxels per chunk
    constexpr int LDK = BP + 2;
    constexpr int RPP = 256 / BP;          // rows per staging pass (8)
    constexpr int APASS = BM / RPP;
    constexpr int BPASS = BN / RPP;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of 8");

    __shared__ float As[2][BM * LDK];
    __shared__ float Bs[2][BN * LDK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WGN) * (16 * TM);
    const int wn0 = (wave % WGN) * (16 * TN);
    const int m0 = (blockIdx.x % tiles_m) * BM;
    const int n0 = (blockIdx.x / tiles_m) * BN;
    const int slice = blockIdx.y;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, prm.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);

    const int pcol = tid % BP;
    const int prow0 = tid / BP;

    // the k-rows (and dY rows) a thread stages are the same in every chunk: decode them once
    int goff[BPASS];        // byte offset of row k inside the gathered tensor (-1: k >= K)
    int gsel[BPASS];        // packed mask shifts  kw | (8+kh)<<8 | (16+kt)<<16
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
        const int k = n0 + prow0 + RPP * j;
        goff[j] = 0;
        gsel[j] = 31 | (31 << 8) | (31 << 16);
        if (k < prm.K) {
            const int c = k / prm.taps;
            const int tap = k - c * prm.taps;
            const int kt = tap / prm.kHW;
            const int rr = tap - kt * prm.kHW;
            const int kh = rr / prm.kW;
            const int kw = rr - kh * prm.kW;
            goff[j] = 4 * (c * prm.gS + kt * prm.gHW + kh * prm.gW + kw);
            gsel[j] = kw | ((8 + kh) << 8) | ((16 + kt) << 16);
        }
    }

    const int chunk_begin = slice * prm.chunks_per_slice;
    int chunk_end = chunk_begin + prm.chunks_per_slice;
    const int total_chunks = (prm.P + BP - 1) / BP;
    if (chunk_end > total_chunks) chunk_end = total_chunks;

    float areg[APASS], breg[BPASS];

    auto load_chunk = [&](int chunk) {
        const int p = chunk * BP + pcol;
        unsigned dy_base = 0xFFFFFFFFu;
        int x_base = 0;
        unsigned vmask = 0;
        if (p < prm.P) {
            const int n = p / prm.oS;
            int r = p - n * prm.oS;
            dy_base = 4u * (unsigned)(n * prm.M * prm.oS + r);
            const int ot = r / prm.oHW;
            r -= ot * prm.oHW;
            const int oh = r / prm.oW;
            const int ow = r - oh * prm.oW;
            const int t0 = ot * prm.sT - prm.pT, h0 = oh * prm.sH - prm.pH, w0 = ow * prm.sW - prm.pW;
            x_base = 4 * (n * prm.gC * prm.gS + t0 * prm.gHW + h0 * prm.gW + w0);
            for (int k = 0; k < prm.kW; ++k) vmask |= ((unsigned)(w0 + k) < (unsigned)prm.gW) << k;
            for (int k = 0; k < prm.kH; ++k) vmask |= ((unsigned)(h0 + k) < (unsigned)prm.gH) << (8 + k);
            for (int k = 0; k < prm.kT; ++k) vmask |= ((unsigned)(t0 + k) < (unsigned)prm.gT) << (16 + k);
        }
        const unsigned row_bytes = 4u * (unsigned)prm.oS;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            const int row = m0 + prow0 + RPP * j;
            // rows >= M land beyond the buffer only for the last clip; mask them explicitly
            const unsigned off = (row < prm.M) ? dy_base + (unsigned)row * row_bytes : 0xFFFFFFFFu;
            areg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc, (int)(off | (dy_base == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u)), 0, 0));
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int sh = gsel[j];
            const unsigned ok = (vmask >> (sh & 31)) & (vmask >> ((sh >> 8) & 31)) & (vmask >> ((sh >> 16) & 31)) & 1u;
            const unsigned off = (unsigned)(x_base + goff[j]) | (ok - 1u);
            breg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (int)off, 0, 0));
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < APASS; ++j) As[buf][(prow0 + RPP * j) * LDK + pcol] = areg[j];
#pragma unroll
        for (int j = 0; j < BPASS; ++j) Bs[buf][(prow0 + RPP * j) * LDK + pcol] = breg[j];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (chunk_begin < chunk_end) {
        load_chunk(chunk_begin);
        store_chunk(0);
    }
    __syncthreads();

    const int frag_k = lane >> 4;
    const int frag_r = lane & 15;
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
        const int cur = (ch - chunk_begin) & 1;
        const bool more = (ch + 1) < chunk_end;
        if (more) load_chunk(ch + 1);
        const float* as = &As[cur][0];
        const float* bs = &Bs[cur][0];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float a[BP / 8][TM], b[BP / 8][TN];
#pragma unroll
            for (int kk = 0; kk < BP / 8; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[kk][i] = as[(wm0 + 16 * i + frag_r) * LDK + (half * (BP / 8) + kk) * 4 + frag_k];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[kk][j] = bs[(wn0 + 16 * j + frag_r) * LDK + (half * (BP / 8) + kk) * 4 + frag_k];
            }
            __builtin_amdgcn_sched_barrier(0);      // fragment burst stays ahead of the MFMA chain
#pragma unroll
            for (int kk = 0; kk < BP / 8; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // partial slab of this slice: OUT[slice][m][k]
    float* out = OUT + (size_t)slice * prm.M * prm.K;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int k = n0 + wn0 + 16 * j + frag_r;
        if (k >= prm.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm0 + 16 * i + 4 * frag_k + r;
                if (m < prm.M) out[(size_t)m * prm.K + k] = acc[i][j][r];
            }
        }
    }
}

// dw[i] = sum_s slab[s][i].  A block owns 32 elements x 8 slice groups (group g adds slices
// g, g+8, ... in order; the 8 partials are then added in group order): fixed order, deterministic,
// and the few thousand elements of a stem gradient still spread over the whole chip.
__global__ __launch_bounds__(256) void slab_sum_generic_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                       long n, int slices) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (long i0 = (long)blockIdx.x * 32; i0 < n; i0 += (long)gridDim.x * 32) {
        const long i = i0 + e;
        float s = 0.f;
        if (i < n)
            for (int k = g; k < slices; k += 8) s += slabs[(size_t)k * n + i];
        part[g][e] = s;
        __syncthreads();
        if (g == 0 && i < n) {
            float t = part[0][e];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += part[q][e];
            out[i] = t;
        }
        __syncthreads();
    }
}

// ================================================================================================
// The R(2+1)D stem's weight gradient (resnet.py:170: Conv3d(3, 45, (1,7,7), stride (1,2,2), padding (0,3,3))) with its operands as ROWS.
// It is the last kernel of a training step and runs alone (its dY is the last gradient the backward produces), so its time is on the
// critical path in full; the generic kernel above gathers every (ci, kh, kw) element of a voxel on its own (4-byte loads, a mask test
// each) and reaches 28 TFLOP/s.  Here a chunk is one OUTPUT ROW (n, t, h'): k = w' (Wo / 4 MFMA steps),
//     dW[co][ci][kh][kw] += sum_w' dY[co][n,t,h',w'] * X[ci][n,t, 2h'+kh-3, 2w'+kw-3]
//   * A = the row of dY for every output channel ([48][Wo], 16-byte LDS-DMAs, four rows per instruction);
//   * B = the seven input rows 2h'-3 .. 2h'+3 of the three channels, kept in a ring of 10 row slots per channel (consecutive output rows
//     share five of them: two new rows per channel and chunk), each row one 16-byte LDS-DMA instruction into a zero-padded slot; a B
//     fragment is read at stride 2 along the row (address 2w' + kw + 1), the row out of the image is a DMA that reads zeros;
//   * columns = 3 x 64 (a channel's 49 taps padded to 64), wave w owns columns 16w .. 16w+15 of each channel, 3 x 3 accumulator tiles;
//   * a workgroup walks `rows` consecutive output rows of one frame and writes its partial dW in the weight tensor's own layout; the
//     partials are added in a fixed order by slab_sum_generic_kernel.
struct StemWgradParams {
    int N, T, Ho, Wo, Hi, Wi, Cout;
    int rows, parts;                 // output rows per workgroup, workgroups per frame
    unsigned x_bytes, dy_bytes;
};

__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemWgradParams prm, const float* __restrict__ X, const float* __restrict__ DY,
                                                         float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int XS = 10, XP = 264;            // ring slots per channel, floats per slot (64 lanes x 16 B + 8: consecutive slots 8 banks apart)
    constexpr int AB = 260;                     // floats per block of four dY rows (one DMA instruction + 4)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    __shared__ __attribute__((aligned(16))) float xs[3 * XS * XP];
    __shared__ __attribute__((aligned(16))) float as_[2][12 * AB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    const int frame = blockIdx.x / prm.parts, part = blockIdx.x - frame * prm.parts;
    const int n = frame / prm.T, t = frame - n * prm.T;
    const int h0 = part * prm.rows, h1 = min(prm.Ho, h0 + prm.rows);
    const int pa = prm.Wo >> 2, px = prm.Wi >> 2;          // 16-byte pieces per dY row / X row

    for (int i = tid; i < 3 * XS * XP; i += 256) xs[i] = 0.f;       // (the left pad of every slot stays zero for good)
    __syncthreads();

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, prm.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);
    // X row rr (= input row + 3) of channel ci = wave: lanes 0 .. px-1 carry the row, the others read zeros (the slot's right pad)
    const long x_plane = ((long)(n * 3 + wave) * prm.T + t) * prm.Hi * prm.Wi;
    auto issue_x = [&](int rr) {
        if (wave >= 3) return;
        const int xr = rr - 3;
        const unsigned off = (lane < px && xr >= 0 && xr < prm.Hi) ? (unsigned)(4 * (x_plane + (long)xr * prm.Wi + 4 * lane)) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t)(xs + (wave * XS + rr % XS) * XP + 4), 16, (int)off, 0, 0, 0);
    };
    // dY row h' of output channels 4b .. 4b+3, b = wave, wave + 4, wave + 8: lane = (row lane / pa, piece lane % pa)
    const int a_row = lane / pa, a_piece = lane - a_row * pa;
    auto issue_dy = [&](int hh, int buf) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int b = wave + 4 * j, co = 4 * b + a_row;
            const unsigned off = (a_row < 4 && co < prm.Cout)
                ? (unsigned)(4 * ((((long)(n * prm.Cout + co) * prm.T + t) * prm.Ho + hh) * prm.Wo + 4 * a_piece)) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dy_rsrc, (lds_ptr_t)(&as_[buf][b * AB]), 16, (int)off, 0, 0, 0);
        }
    };

    // this lane's fragment columns: q = 16 * wave + r16 of every channel -> tap (kh, kw) (the padding columns 49..63 read tap 0 and are dropped)
    const int q = 16 * wave + r16, qq = q < 49 ? q : 0, kh = qq / 7, kw = qq - 7 * kh;
    int a_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int co = 16 * i + r16;
        a_off[i] = (co >> 2) * AB + (co & 3) * (4 * pa) + g;
    }
    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (h0 < h1) {
        for (int r = 0; r < 7; ++r) issue_x(2 * h0 + r);
        issue_dy(h0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int ksteps = prm.Wo >> 2;
    for (int hh = h0; hh < h1; ++hh) {
        const int cur = (hh - h0) & 1;
        if (hh + 1 < h1) {
            issue_x(2 * hh + 7);
            issue_x(2 * hh + 8);
            issue_dy(hh + 1, cur ^ 1);
        }
        const float* as = as_[cur];
        int b_off[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) b_off[j] = (j * XS + (2 * hh + kh) % XS) * XP + kw + 1 + 2 * g;
        float a[2][3], b[2][3];
        auto fetch = [&](int s, int slot) {
#pragma unroll
            for (int i = 0; i < 3; ++i) a[slot][i] = as[a_off[i] + 4 * s];
#pragma unroll
            for (int j = 0; j < 3; ++j) b[slot][j] = xs[b_off[j] + 8 * s];
        };
        fetch(0, 0);
        for (int s = 0; s < ksteps; s += 2) {                       // (two steps per trip: the fragment slots stay compile-time)
            if (s + 1 < ksteps) fetch(s + 1, 1);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
            if (s + 1 < ksteps) {
                if (s + 2 < ksteps) fetch(s + 2, 0);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // partial dW of this workgroup, in the weight tensor's layout [co][ci][kh][kw]; acc[i][j][r]: co = 16 i + 4 g + r, column q of channel j
    float* out = OUT + (size_t)blockIdx.x * prm.Cout * 147;
    if (q < 49) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * i + 4 * g + r;
                if (co < prm.Cout) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) out[co * 147 + j * 49 + q] = acc[i][j][r];
                }
            }
    }
#endif
}

static bool stem_wgrad_shape(const zsv_conv_desc* d) {
    if (ZSV_KNOB(NO_STEM_WGRAD)) return false;
    if (d->Cin != 3 || d->kT != 1 || d->sT != 1 || d->pT != 0 || d->kH != 7 || d->kW != 7 || d->sH != 2 || d->sW != 2 || d->pH != 3 || d->pW != 3)
        return false;
    if (d->Hi % 2 != 0 || d->Wi % 4 != 0 || d->Wo != d->Wi / 2 || d->Ho != d->Hi / 2 || d->Wo % 4 != 0) return false;
    if (d->Wo > 64 || d->Cout > 48 || d->Cout < 1) return false;          // four dY rows per DMA instruction: 4 * Wo / 4 <= 64 lanes
    if ((long)d->N * d->Cout * d->To * d->Ho * d->Wo >= (1L << 29) || (long)d->N * 3 * d->Ti * d->Hi * d->Wi >= (1L << 29)) return false;
    return true;
}
// workgroups per frame: ~1400 in all (2.75 rounds of the 512 resident ones measured best at 352 frames: 0.24 ms against 0.25-0.28 for 1056, 2112,
// 2816), at least 8 output rows each, and an even split of the rows where one is near
static int stem_wgrad_parts(const zsv_conv_desc* d) {
    const long frames = (long)d->N * d->To;
    long target = 1408;
    if (const char* e = ZSV_KNOB(STEM_WGRAD_WGS)) target = atol(e) > 0 ? atol(e) : target;
    long parts = (target + frames - 1) / frames;
    const long maxp = (d->Ho + 7) / 8;
    if (parts > maxp) parts = maxp;
    if (parts < 1) parts = 1;
    for (long q = parts; q * 2 > parts && q >= 1; --q)
        if (d->Ho % q == 0) { parts = q; break; }
    const long rows = (d->Ho + parts - 1) / parts;
    return (int)((d->Ho + rows - 1) / rows);
}

// first level of a two-level slab sum: block row c adds slabs [c * per, (c + 1) * per) into part[c][i] (fixed order)
__global__ __launch_bounds__(256) void slab_sum_chunks_kernel(const float* __restrict__ slabs, float* __restrict__ part, long n, int slices,
                                                              int per) {
    const int c = blockIdx.y, k0 = c * per, k1 = min(slices, k0 + per);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = k0; k < k1; ++k) s += slabs[(size_t)k * n + i];
        part[(size_t)c * n + i] = s;
    }
}

struct WgradGenPlan {
    int cfg;        // 0: 128x128, 1: 64x128 (small Cout), 2: 144x64
    int tiles_m, tiles_n, slices, chunks_per_slice;
};

static WgradGenPlan wgrad_gen_plan(const zsv_conv_desc* d) {
    WgradGenPlan pl;
    const int M = d->Cout;
    const int K = d->Cin * d->kT * d->kH * d->kW;
    const long P = (long)d->N * d->To * d->Ho * d->Wo;
    int bm, bn;
    if (M <= 64) { pl.cfg = 1; bm = 64; bn = 128; }
    else if (M % 144 == 0 || (M > 128 && M <= 144)) { pl.cfg = 2; bm = 144; bn = 64; }
    else { pl.cfg = 0; bm = 128; bn = 128; }
    pl.tiles_m = (M + bm - 1) / bm;
    pl.tiles_n = (K + bn - 1) / bn;
    const long chunks = (P + 31) / 32;
    const long tiles = (long)pl.tiles_m * pl.tiles_n;
    long slices = (1536 + tiles - 1) / tiles;            // aim at ~6 workgroups per CU
    long max_slices = (chunks + 15) / 16;                // at least 16 chunks (512 voxels) per slice
    if (max_slices < 1) max_slices = 1;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    if (slices > 1024) slices = 1024;
    pl.chunks_per_slice = (int)((chunks + slices - 1) / slices);
    pl.slices = (int)((chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice);
    return pl;
}


static size_t wgrad_gen_plan_bytes(const zsv_conv_desc* d) {
    const WgradGenPlan pl = wgrad_gen_plan(d);
    if (pl.slices <= 1) return 0;
    return (size_t)pl.slices * d->Cout * d->Cin * d->kT * d->kH * d->kW * sizeof(float);
}
// slabs of the workgroups + the 32 partial sums of the first reduction level
static size_t stem_wgrad_bytes(const zsv_conv_desc* d) { return ((size_t)d->N * d->To * stem_wgrad_parts(d) + 32) * d->Cout * 147 * sizeof(float); }

size_t wgrad_generic_workspace_bytes(const zsv_conv_desc* d) {
    const size_t gen = wgrad_gen_plan_bytes(d);
    if (!stem_wgrad_shape(d)) return gen;
    const size_t stem = stem_wgrad_bytes(d);            // (the generic kernel still serves unaligned tensors of this shape)
    return stem > gen ? stem : gen;
}

int wgrad_generic(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                  size_t workspace_bytes, hipStream_t stream) {
    if (stem_wgrad_shape(d) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
        if (!workspace || workspace_bytes < stem_wgrad_bytes(d)) return ZSV_E_WORKSPACE;
        StemWgradParams p;
        p.N = d->N; p.T = d->To; p.Ho = d->Ho; p.Wo = d->Wo; p.Hi = d->Hi; p.Wi = d->Wi; p.Cout = d->Cout;
        p.parts = stem_wgrad_parts(d);
        p.rows = (d->Ho + p.parts - 1) / p.parts;
        p.x_bytes = 4u * (unsigned)((long)d->N * 3 * d->Ti * d->Hi * d->Wi);
        p.dy_bytes = 4u * (unsigned)((long)d->N * d->Cout * d->To * d->Ho * d->Wo);
        const long wgs = (long)d->N * d->To * p.parts;
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, p, x, dy, (float*)workspace);
        if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
        const long n = (long)d->Cout * 147;
        const long blocks = (n + 31) / 32;
        if (wgs <= 64) {
            hipLaunchKernelGGL(slab_sum_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)workspace, dw, n, (int)wgs);
            return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
        }
        // two levels, both in a fixed order: 32 chunks of slabs, then the 32 partial sums
        float* part = (float*)workspace + (size_t)wgs * n;
        const int per = (int)((wgs + 31) / 32), chunks = (int)((wgs + per - 1) / per);
        hipLaunchKernelGGL(slab_sum_chunks_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)chunks), dim3(256), 0, stream, (const float*)workspace,
                           part, n, (int)wgs, per);
        if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
        hipLaunchKernelGGL(slab_sum_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)part, dw, n, chunks);
        return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
    }
    const WgradGenPlan pl = wgrad_gen_plan(d);
    const size_t need = wgrad_gen_plan_bytes(d);
    if (need > 0 && (!workspace || workspace_bytes < need)) return ZSV_E_WORKSPACE;
    WgradGenParams p;
    p.M = d->Cout;
    p.taps = d->kT * d->kH * d->kW;
    p.K = d->Cin * p.taps;
    p.P = d->N * d->To * d->Ho * d->Wo;
    p.kHW = d->kH * d->kW; p.kW = d->kW; p.kH = d->kH; p.kT = d->kT;
    p.oS = d->To * d->Ho * d->Wo; p.oHW = d->Ho * d->Wo; p.oW = d->Wo;
    p.gC = d->Cin; p.gT = d->Ti; p.gH = d->Hi; p.gW = d->Wi;
    p.gS = d->Ti * d->Hi * d->Wi; p.gHW = d->Hi * d->Wi;
    p.sT = d->sT; p.sH = d->sH; p.sW = d->sW; p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
    p.chunks_per_slice = pl.chunks_per_slice;
    p.x_bytes = 4u * (unsigned)((long)d->N * d->Cin * p.gS);
    p.dy_bytes = 4u * (unsigned)((long)d->N * d->Cout * p.oS);
    float* out = pl.slices > 1 ? (float*)workspace : dw;
    const dim3 grid((unsigned)(pl.tiles_m * pl.tiles_n), (unsigned)pl.slices);
    switch (pl.cfg) {
        case 0: hipLaunchKernelGGL((conv_wgrad_generic_kernel<4, 4, 2, 2>), grid, dim3(256), 0, stream, p, x, dy, out, pl.tiles_m); break;
        case 1: hipLaunchKernelGGL((conv_wgrad_generic_kernel<4, 2, 1, 4>), grid, dim3(256), 0, stream, p, x, dy, out, pl.tiles_m); break;
        default: hipLaunchKernelGGL((conv_wgrad_generic_kernel<9, 1, 1, 4>), grid, dim3(256), 0, stream, p, x, dy, out, pl.tiles_m); break;
    }
    if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
    if (pl.slices > 1) {
        const long n = (long)p.M * p.K;
        long blocks = (n + 31) / 32;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(slab_sum_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)workspace, dw, n, pl.slices);
        if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
    }
    return ZSV_OK;
}

}  // namespace zsv
