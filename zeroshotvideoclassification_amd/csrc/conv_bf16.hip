// conv_bf16.hip -- bf16 inference convolution (BASELINE config 5: 32-frame bf16 eval loop).
//
// Forward-only Conv3d with the eval-mode BatchNorm folded in (main.py:229 `model.eval()`):
//     y = relu?( conv3d(x, w * scale[cout]) + shift[cout] (+ residual) )
// which is what Conv3d -> BatchNorm3d(eval) -> ReLU (resnet.py:40-52,94-98) and the block tail
// `out += residual; relu` (resnet.py:110-111) compute.  Products are bf16 x bf16 accumulated in
// fp32 on v_mfma_f32_16x16x32_bf16; activations travel between layers in bf16.
//
// Layout (channels-last, so that the MFMA's 8 consecutive k elements are 16 contiguous bytes):
//     activations  [N][T][H][W][Cp]  bf16, Cp = Cin rounded up to 32, pad channels are zero
//     weights      Wp[q][Mp][32]     bf16, q = tap * (Cp/32) + chunk, rows padded to the row tile
//     blob         Wp followed by Mp fp32 shifts
// A clip (Cin = 3) uses the "folded" form: pixels are stored [N][T][Hp][Wp][4] with the H/W zero
// padding materialised, and one K chunk of 32 is 8 consecutive pixels x 4 channels of one (kt, kh)
// row, i.e. kw is folded into the chunk (taps = kT*kH, weights at k = 4*kw + c).
//
// Implicit GEMM: rows = produced channels, columns = output voxels, K = taps x Cp in chunks of 32.
// One workgroup = 4 waves computes a (16*TM*WGM) x (16*TN*WGN) tile; per K chunk the A rows
// (BM x 64 B, contiguous in Wp) and the B rows (one 64-byte channel segment per voxel, zero line for
// padding taps) are staged through registers into a double-buffered, XOR-swizzled LDS image and read
// back as ds_read_b128 fragments.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <type_traits>

#include "conv_params.h"
#include "zsv_common.h"
#include "zsv_hip.h"
#include "knobs.h"

namespace zsv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ u32x4 zsv_zero_line[8];      // 128 zero bytes: what a padding tap reads

struct Bf16Params {
    int M, Mp, CoutP;          // produced channels, packed rows (row tiles x BM), output channel pitch
    int nq, nchunk;            // K chunks of 32: total and per tap
    int kT, kH, kW;            // tap grid (kW = 1 in the folded form)
    int sT, sH, sW;            // input elements per t / h / w step
    long sN;                   // input elements per clip
    int Ti, Hi, Wi;            // input extents (validity of a tap)
    int ToHoWo, HoWo, Wo;
    int strT, strH, strW, pT, pH, pW;
    int P;                     // output voxels N*To*Ho*Wo
    int tiles_m, tiles_n;
    int relu;
    int cc_outer;              // shared-image kernel: walk the K chunks outermost (image rows of one chunk stay in L2 across the kh groups)
    float* stat;               // != nullptr: per column tile the sums and sums of squares of the STORED (bf16-rounded) values, [tiles_n][2][CoutP]
};

// 16-byte slot swizzle of a 64-byte LDS row: ds_read_b128 serves lanes in the groups
// {0-3,12-15,20-27},{4-11,16-19,28-31},... (MI355X_MICROARCH.md, LDS table).  With slot ^= swz(row)
// the 16 lanes of every group fall on 16 distinct slots of the 256-byte bank row, for a fragment of 16
// consecutive rows starting at ANY row (checked exhaustively) -- the shifted reads of
// conv_bf16_same_kernel need that.
__host__ __device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }

// Channel (relative to the row tile) that packed row `rho` of a BM-row tile holds.  Row-block pairs
// are interleaved so that in the epilogue a lane's 4 + 4 accumulator rows of blocks (2k, 2k+1) are 8
// consecutive channels = one 16-byte store; an unpaired last block keeps the natural order.
__host__ __device__ __forceinline__ int tile_channel(int rho, int bm) {
    const int i = rho >> 4, r = rho & 15, blocks = bm >> 4;
    if (i < (blocks & ~1)) return 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3);
    return rho;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Fragment reads and their waits are written as asm: with an LDS-DMA in flight hipcc only ever emits
// `s_waitcnt lgkmcnt(0)`, which would expose the latency of the newest ds_read at every use; here the
// waits are counted (reads return in issue order) and tied to the registers they release.
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(bf16x8& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}

__device__ __forceinline__ void add_bf16x8(f32x4& lo, f32x4& hi, u32x4 r) {
    lo[0] += __builtin_bit_cast(float, r[0] << 16); lo[1] += __builtin_bit_cast(float, r[0] & 0xffff0000u);
    lo[2] += __builtin_bit_cast(float, r[1] << 16); lo[3] += __builtin_bit_cast(float, r[1] & 0xffff0000u);
    hi[0] += __builtin_bit_cast(float, r[2] << 16); hi[1] += __builtin_bit_cast(float, r[2] & 0xffff0000u);
    hi[2] += __builtin_bit_cast(float, r[3] << 16); hi[3] += __builtin_bit_cast(float, r[3] & 0xffff0000u);
}

// One K chunk of a wave's tile: B0..B3, A0, A1 fragments up front, then A(i+2) under the MFMAs of
// A(i); reads return in issue order, so before using A(i) at most min(2, TM-1-i) younger reads may
// still be out.  MASKED: column block j is zeroed unless keep_j != 0 (a tap outside the input).
template <int TM, int TN, bool MASKED>
__device__ __forceinline__ void mfma_step(f32x4 (&acc)[TM][TN], unsigned sa, unsigned sb, const unsigned (&keep)[TN]) {
    static_assert(TN == 4 || TN == 8, "fragment schedule is written for 4 or 8 column blocks");
    bf16x8 bf[TN], af[3];
    lds_read128<0>(bf[0], sb);
    lds_read128<1024>(bf[1], sb);
    lds_read128<2048>(bf[2], sb);
    lds_read128<3072>(bf[3], sb);
    if constexpr (TN == 8) {
        lds_read128<4096>(bf[4], sb);
        lds_read128<5120>(bf[5], sb);
        lds_read128<6144>(bf[6], sb);
        lds_read128<7168>(bf[7], sb);
    }
    lds_read128<0>(af[0], sa);
    if (TM > 1) lds_read128<1024>(af[1], sa);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (i + 2 < TM) lds_read128<0>(af[(i + 2) % 3], sa + (i + 2) * 1024);
        if (i == 0) {
            if (TM > 2) lds_wait<2>(af[0], bf[0], bf[1], bf[2], bf[3]);
            else if (TM > 1) lds_wait<1>(af[0], bf[0], bf[1], bf[2], bf[3]);
            else lds_wait<0>(af[0], bf[0], bf[1], bf[2], bf[3]);
            if constexpr (TN == 8)       // (issued before af[0]: returned by the wait above; tie the registers to it)
                asm volatile("" : "+v"(bf[4]), "+v"(bf[5]), "+v"(bf[6]), "+v"(bf[7]));
            if (MASKED) {
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = keep[j] ? bf[j] : z;
            }
        } else if (i + 2 < TM) lds_wait<2>(af[i % 3]);
        else if (i + 1 < TM) lds_wait<1>(af[i % 3]);
        else lds_wait<0>(af[i % 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % 3], bf[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// + shift (+ residual), relu, bf16, channels-last stores (16 bytes per lane for paired row blocks);
// `sh` = this row tile's shifts in LDS.
// Column c of the tile is voxel n0 + c, or -- hb_shift > 0, the (frames x positions) tiles of
// conv_bf16_tsame_kernel -- voxel n0 + (c >> hb_shift) * frame_stride + (c & (HB - 1)).
// prm.stat != nullptr (training forward in front of a BatchNorm: no residual, no ReLU): the tile's per-channel sum and sum of squares of
// the values as STORED (rounded to bf16 -- what the BatchNorm's own statistics pass would read back) go to
// stat[tn][0 / 1][m0 + channel]: in-lane over the wave's column blocks, 16-lane DPP sums over a block's columns, the waves of a row
// through `scratch` (the stage area of LDS, free after the K loop) in wave order -- fixed order, reproducible.
template <int TM, int TN, int BM, int BN>
__device__ __forceinline__ void epilogue(const Bf16Params& prm, f32x4 (&acc)[TM][TN], const float* sh,
                                         const __bf16* __restrict__ R, __bf16* __restrict__ Y, int m0, int n0, int tm,
                                         int wm, int wn, int tid, int hb_shift = 0, int frame_stride = 0, float* scratch = nullptr,
                                         int tn = 0, int wgn = 4) {
    const int lane = tid & 63;
    const bool stats = prm.stat != nullptr;
    float ssum[TM][4], ssq[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[i][e] = ssq[i][e] = 0.f;
    auto voxel = [&](int c) { return hb_shift ? n0 + (c >> hb_shift) * frame_stride + (c & ((1 << hb_shift) - 1)) : n0 + c; };
    const float floor_ = prm.relu ? 0.f : -__builtin_inff();
    const int g = lane >> 4;
    const int ch_t = wm * TM * 16;                   // this wave's first channel inside the row tile
    constexpr int NPAIR = TM / 2;
    auto finish = [&](auto has_res) {
        constexpr bool HAS_RES = decltype(has_res)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = voxel((wn * TN + j) * 16 + (lane & 15));
            const bool cv = col < prm.P;
            const size_t row_off = (size_t)(cv ? col : 0) * prm.CoutP + m0;
            u32x4 res[NPAIR > 0 ? NPAIR : 1];
            u32x2 res_odd = u32x2{0u, 0u};
            if (HAS_RES) {
#pragma unroll
                for (int k = 0; k < NPAIR; ++k) {
                    const int ch = ch_t + 32 * k + 8 * g;
                    res[k] = *(const u32x4*)(R + row_off + (m0 + ch < prm.CoutP ? ch : 0));
                }
                if (TM & 1) {
                    const int ch = ch_t + 16 * (TM - 1) + 4 * g;
                    res_odd = *(const u32x2*)(R + row_off + (m0 + ch < prm.CoutP ? ch : 0));
                }
            }
#pragma unroll
            for (int k = 0; k < NPAIR; ++k) {
                const int ch = ch_t + 32 * k + 8 * g;
                f32x4 lo = acc[2 * k][j] + *(const f32x4*)(sh + ch);
                f32x4 hi = acc[2 * k + 1][j] + *(const f32x4*)(sh + ch + 4);
                if (HAS_RES) add_bf16x8(lo, hi, res[k]);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (__bf16)fmaxf(lo[e], floor_);
                    o[4 + e] = (__bf16)fmaxf(hi[e], floor_);
                }
                if (cv && m0 + ch < prm.CoutP) *(bf16x8*)(Y + row_off + ch) = o;
                if (stats && cv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = (float)o[e], b = (float)o[4 + e];
                        ssum[2 * k][e] += a; ssq[2 * k][e] = fmaf(a, a, ssq[2 * k][e]);
                        ssum[2 * k + 1][e] += b; ssq[2 * k + 1][e] = fmaf(b, b, ssq[2 * k + 1][e]);
                    }
                }
            }
            if (TM & 1) {                             // unpaired last row block: 4 channels per lane
                const int ch = ch_t + 16 * (TM - 1) + 4 * g;
                f32x4 v = acc[TM - 1][j] + *(const f32x4*)(sh + ch);
                if (HAS_RES) {
                    v[0] += __builtin_bit_cast(float, res_odd[0] << 16);
                    v[1] += __builtin_bit_cast(float, res_odd[0] & 0xffff0000u);
                    v[2] += __builtin_bit_cast(float, res_odd[1] << 16);
                    v[3] += __builtin_bit_cast(float, res_odd[1] & 0xffff0000u);
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(v[e], floor_);
                if (cv && m0 + ch < prm.CoutP) *(bf16x4*)(Y + row_off + ch) = o;
                if (stats && cv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = (float)o[e];
                        ssum[TM - 1][e] += a; ssq[TM - 1][e] = fmaf(a, a, ssq[TM - 1][e]);
                    }
                }
            }
        }
    };
    if (R != nullptr) finish(std::true_type{});
    else finish(std::false_type{});
    if (stats) {
        // tile-relative channel of (row block i, element e) of this lane: the paired blocks interleave (tile_channel)
        auto row16 = [](float v) {                   // sum over the 16 lanes of a DPP row (they hold the block's 16 columns)
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
            return v;
        };
        __syncthreads();                              // every wave is past its last fragment read: the stage area is free
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bool paired = i < 2 * NPAIR;
            const int chb = paired ? ch_t + 32 * (i >> 1) + 8 * g + 4 * (i & 1) : ch_t + 16 * (TM - 1) + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = row16(ssum[i][e]), b = row16(ssq[i][e]);
                if ((lane & 15) == 0) {
                    scratch[(wn * BM + chb + e) * 2] = a;
                    scratch[(wn * BM + chb + e) * 2 + 1] = b;
                }
            }
        }
        __syncthreads();
        if (tid < BM && m0 + tid < prm.CoutP) {
            float a = 0.f, b = 0.f;
            for (int w = 0; w < wgn; ++w) { a += scratch[(w * BM + tid) * 2]; b += scratch[(w * BM + tid) * 2 + 1]; }
            float* out = prm.stat + (size_t)tn * 2 * prm.CoutP + m0 + tid;
            out[0] = a;
            out[prm.CoutP] = b;
        }
    }
    // channels between the last packed row and the pitch (e.g. 144 -> 160) stay zero
    const int covered = prm.tiles_m * BM;
    if (tm == prm.tiles_m - 1 && covered < prm.CoutP) {
        const int per_col = (prm.CoutP - covered) >> 2;            // 8-byte pieces per voxel
        for (int idx = tid; idx < BN * per_col; idx += 256) {
            const int c = idx / per_col, k = idx - c * per_col;
            const int col = voxel(c);
            if (col < prm.P) *(u32x2*)(Y + (size_t)col * prm.CoutP + covered + 4 * k) = u32x2{0u, 0u};
        }
    }
}

// Pipeline: a ring of 3 LDS stages filled by LDS-DMA (global_load_lds_dwordx4, no staging registers)
// two K chunks ahead of the MFMAs; a stage is 1-KiB pieces (16 rows x 64 B) dealt round-robin to the 4
// waves, each lane fetching the 16 bytes that belong at its (row, swizzled slot).  Every wave issues
// the same number of DMAs per stage (NPW), so `s_waitcnt vmcnt(NPW)` retires exactly the older stage.
template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(Bf16Params prm, const __bf16* __restrict__ X,
                                                           const __bf16* __restrict__ Wp,
                                                           const float* __restrict__ shift,
                                                           const __bf16* __restrict__ R, __bf16* __restrict__ Y) {
#if defined(__HIP_DEVICE_COMPILE__)     // (address_space(3) casts: the host pass would drop the stub)
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(WGM == 1 || (TM % 2) == 0, "row-block pairs must not straddle waves");
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int NA_P = BM / 16, NB_P = BN / 16;    // 1-KiB pieces per stage
    static_assert(NA_P >= 4 && NB_P % 4 == 0, "piece distribution");
    constexpr int NAW = (NA_P + 3) / 4, NBW = NB_P / 4;
    constexpr int NPW = NAW + NBW;                   // DMAs per wave per stage
    constexpr int STAGE = (BM + BN) * 64;
    constexpr int SHIFT_AT = 3 * STAGE;              // 1 KiB: this row tile's fp32 shifts
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int tile = xcd_tile(gridDim.x, blockIdx.x);
    const int tm = tile % prm.tiles_m, tn = tile / prm.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA assignment ------------------------------------------------------------------------
    // lane l of a piece fills row l/4, slot l%4 of the LDS image <- source slot (l%4) ^ swz(row)
    const int srcslot = ((lane & 3) ^ swz(lane >> 2)) * 8;
    int a_off[NAW], a_dst[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        int pa = wave + 4 * k;
        if (pa >= NA_P) pa -= 4;                      // surplus slot: repeat this wave's previous piece
        a_off[k] = (pa * 16 + (lane >> 2)) * 32 + srcslot;
        a_dst[k] = pa * 1024;
    }
    int b_base[NBW], b_dst[NBW];
    unsigned b_mask[NBW];                            // bit tap = the tap lies inside the input
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int pb = wave + 4 * k;
        const int p = n0 + pb * 16 + (lane >> 2);
        b_dst[k] = BM * 64 + pb * 1024;
        b_mask[k] = 0;
        b_base[k] = 0;
        if (p < prm.P) {
            const int n = p / prm.ToHoWo;
            int rem = p - n * prm.ToHoWo;
            const int to = rem / prm.HoWo;
            rem -= to * prm.HoWo;
            const int ho = rem / prm.Wo, wo = rem - ho * prm.Wo;
            const int t0 = to * prm.strT - prm.pT, h0 = ho * prm.strH - prm.pH, w0 = wo * prm.strW - prm.pW;
            b_base[k] = (int)(n * prm.sN) + t0 * prm.sT + h0 * prm.sH + w0 * prm.sW + srcslot;
            unsigned m = 0;
            int tap = 0;
            for (int a = 0; a < prm.kT; ++a)
                for (int b = 0; b < prm.kH; ++b)
                    for (int c = 0; c < prm.kW; ++c, ++tap)
                        m |= (unsigned)((unsigned)(t0 + a) < (unsigned)prm.Ti && (unsigned)(h0 + b) < (unsigned)prm.Hi &&
                                        (unsigned)(w0 + c) < (unsigned)prm.Wi) << tap;
            b_mask[k] = m;
        }
    }
    const __bf16* wq = Wp + (size_t)m0 * 32;         // chunk q lives at + q*Mp*32
    const size_t wq_step = (size_t)prm.Mp * 32;
    const __bf16* zero = (const __bf16*)zsv_zero_line;

    // uniform walk over (tap, chunk)
    int kt = 0, kh = 0, kw = 0, cc = 0, tap_i = 0;
    auto issue = [&](int buf) {
        unsigned char* base = lds + buf * STAGE;
        const int tapoff = kt * prm.sT + kh * prm.sH + kw * prm.sW + cc * 32;
#pragma unroll
        for (int k = 0; k < NAW; ++k)
            __builtin_amdgcn_global_load_lds(wq + a_off[k], (lds_ptr_t)(base + a_dst[k]), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            const __bf16* src = ((b_mask[k] >> tap_i) & 1u) ? X + (b_base[k] + tapoff) : zero;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(base + b_dst[k]), 16, 0, 0);
        }
        wq += wq_step;
        if (++cc == prm.nchunk) {
            cc = 0;
            ++tap_i;
            if (++kw == prm.kW) {
                kw = 0;
                if (++kh == prm.kH) { kh = 0; ++kt; }
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ swz(lane & 15)) << 4);
    const int a_frag = wm * TM * 16 * 64 + frag_off;
    const int b_frag = BM * 64 + wn * TN * 16 * 64 + frag_off;

    if (wave == 0) {                                  // oldest DMA of wave 0: retired by its first counted wait
        const int l4 = lane < BM / 4 ? lane : BM / 4 - 1;
        __builtin_amdgcn_global_load_lds(shift + m0 + 4 * l4, (lds_ptr_t)(lds + SHIFT_AT), 16, 0, 0);
    }
    issue(0);
    if (prm.nq > 1) {
        issue(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt2 = 2;
    for (int q = 0; q < prm.nq; ++q) {
        const bool ahead = q + 2 < prm.nq;
        if (ahead) issue(nxt2);
        const unsigned no_mask[TN] = {};
        mfma_step<TM, TN, false>(acc, lds_base + cur * STAGE + a_frag, lds_base + cur * STAGE + b_frag, no_mask);
        // stage q+1 must have landed (all but this wave's newest NPW DMAs), for every wave
        if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }

    epilogue<TM, TN, BM, BN>(prm, acc, (const float*)(lds + SHIFT_AT), R, Y, m0, n0, tm, wm, wn, tid, 0, 0, (float*)lds, tn, WGN);
#endif
}

// Stride-1 "same" convolutions with kW = 3 (the 1x3x3 spatial convs of Conv2Plus1D, resnet.py:40-45,
// and 3x3x3, resnet.py:23-30): in the flattened voxel index v = ((n*T + t)*H + h)*W + w a tap is a
// constant shift  d = (kt-pT)*H*W + (kh-pH)*W + (kw-pW)  plus a per-voxel border test.  For one
// (kt, kh) the three kw taps read rows v+d, v+d+1, v+d+2: ONE LDS image of BN+2 consecutive input rows
// serves all three (fragments are read at row offsets 0/1/2), and the border test becomes a per-lane
// zeroing of the B fragment instead of a per-row address select.  That cuts the bytes gathered from
// L2 into LDS -- the limiter of the per-tap kernel on these layers -- by 1.7x for Cin = 64.
// LDS: ring of 3 A stages (per tap) + ring of 2 B images (per (kt,kh), chunk) + shifts.
template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void conv_bf16_same_kernel(Bf16Params prm, const __bf16* __restrict__ X,
                                                                const __bf16* __restrict__ Wp,
                                                                const float* __restrict__ shift,
                                                                const __bf16* __restrict__ R, __bf16* __restrict__ Y) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(WGM == 1 || (TM % 2) == 0, "row-block pairs must not straddle waves");
    constexpr int KW = 3;
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int NA_P = BM / 16, NI_P = (BN + KW - 1 + 15) / 16;     // 1-KiB pieces: A stage, B image
    constexpr int NAW = (NA_P + 3) / 4, NIW = (NI_P + 3) / 4;
    constexpr int A_STAGE = BM * 64, IMG = NI_P * 1024;
    constexpr int IMG_AT = 3 * A_STAGE, SHIFT_AT = IMG_AT + 2 * IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int tile = xcd_tile(gridDim.x, blockIdx.x);
    const int tm = tile % prm.tiles_m, tn = tile / prm.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const int srcslot = ((lane & 3) ^ swz(lane >> 2)) * 8;
    int a_off[NAW], a_dst[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        int pa = wave + 4 * k;
        if (pa >= NA_P) pa -= 4;
        a_off[k] = (pa * 16 + (lane >> 2)) * 32 + srcslot;
        a_dst[k] = pa * 1024;
    }
    int i_row[NIW], i_dst[NIW];                      // image row this lane fills, per piece
#pragma unroll
    for (int k = 0; k < NIW; ++k) {
        int pi = wave + 4 * k;
        if (pi >= NI_P) pi -= 4;
        i_row[k] = n0 + pi * 16 + (lane >> 2);
        i_dst[k] = pi * 1024;
    }
    // border masks of this lane's 4 columns: bit tap = the tap reads inside the input.  One decode,
    // then +16 voxels per column block (carry into h, t).
    unsigned mask[TN];
    {
        const int p0 = n0 + wn * TN * 16 + (lane & 15);
        int rem = p0 % prm.ToHoWo;
        int t = rem / prm.HoWo;
        rem -= t * prm.HoWo;
        int h = rem / prm.Wo, w = rem - h * prm.Wo;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            unsigned m = 0;
            if (p0 + 16 * j < prm.P) {
                unsigned mt = 0, mh = 0, mw = 0;
                for (int a = 0; a < prm.kT; ++a) mt |= (unsigned)((unsigned)(t + a - prm.pT) < (unsigned)prm.Ti) << a;
                for (int b = 0; b < prm.kH; ++b) mh |= (unsigned)((unsigned)(h + b - prm.pH) < (unsigned)prm.Hi) << b;
                for (int c = 0; c < KW; ++c) mw |= (unsigned)((unsigned)(w + c - prm.pW) < (unsigned)prm.Wi) << c;
                int tap = 0;
                for (int a = 0; a < prm.kT; ++a)
                    for (int b = 0; b < prm.kH; ++b, tap += KW)
                        if (((mt >> a) & (mh >> b)) & 1u) m |= mw << tap;
            }
            mask[j] = m;
            w += 16;
            while (w >= prm.Wo) {
                w -= prm.Wo;
                if (++h == prm.Hi) { h = 0; if (++t == prm.Ti) t = 0; }
            }
        }
    }
    const __bf16* zero = (const __bf16*)zsv_zero_line;
    const size_t wq_step = (size_t)prm.Mp * 32;
    const __bf16* w_tile = Wp + (size_t)m0 * 32;

    // step = (group (kt,kh), chunk cc, kw); image = (group, cc)
    const int nimg = prm.kT * prm.kH * prm.nchunk;
    const int nsteps = nimg * KW;
    int a_grp = 0, a_cc = 0, a_kw = 0;               // walk of the A issue
    auto issue_a = [&](int buf) {
        const __bf16* wq = w_tile + (size_t)((a_grp * KW + a_kw) * prm.nchunk + a_cc) * wq_step;
        unsigned char* base = lds + buf * A_STAGE;
#pragma unroll
        for (int k = 0; k < NAW; ++k) __builtin_amdgcn_global_load_lds(wq + a_off[k], (lds_ptr_t)(base + a_dst[k]), 16, 0, 0);
        if (++a_kw == KW) {
            a_kw = 0;
            if (prm.cc_outer) {
                if (++a_grp == prm.kT * prm.kH) { a_grp = 0; ++a_cc; }
            } else if (++a_cc == prm.nchunk) { a_cc = 0; ++a_grp; }
        }
    };
    int b_kt = 0, b_kh = 0, b_cc = 0;                // walk of the B image issue
    auto issue_b = [&](int buf) {
        const int shift_rows = (b_kt - prm.pT) * prm.HoWo + (b_kh - prm.pH) * prm.Wo - prm.pW;
        unsigned char* base = lds + IMG_AT + buf * IMG;
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            const int src = i_row[k] + shift_rows;
            const __bf16* ptr = (unsigned)src < (unsigned)prm.P ? X + ((size_t)src * prm.sW + b_cc * 32 + srcslot) : zero;
            __builtin_amdgcn_global_load_lds(ptr, (lds_ptr_t)(base + i_dst[k]), 16, 0, 0);
        }
        if (prm.cc_outer) {
            if (++b_kh == prm.kH) {
                b_kh = 0;
                if (++b_kt == prm.kT) { b_kt = 0; ++b_cc; }
            }
        } else if (++b_cc == prm.nchunk) {
            b_cc = 0;
            if (++b_kh == prm.kH) { b_kh = 0; ++b_kt; }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned a_frag = lds_base + wm * TM * 16 * 64 + (lane & 15) * 64 + (((lane >> 4) ^ swz(lane & 15)) << 4);
    unsigned b_frag[KW];                             // fragment base per kw: rows (l&15) + kw of the column block
#pragma unroll
    for (int c = 0; c < KW; ++c) {
        const int x = (lane & 15) + c;
        b_frag[c] = lds_base + IMG_AT + (wn * TN * 16 + x) * 64 + (((lane >> 4) ^ swz(x)) << 4);
    }

    if (wave == 0) {
        const int l4 = lane < BM / 4 ? lane : BM / 4 - 1;
        __builtin_amdgcn_global_load_lds(shift + m0 + 4 * l4, (lds_ptr_t)(lds + SHIFT_AT), 16, 0, 0);
    }
    issue_b(0);
    issue_a(0);
    issue_a(1);                                      // nsteps >= 3
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW) : "memory");
    __builtin_amdgcn_s_barrier();
    int abuf = 0, abuf2 = 2, tap = 0;
    for (int img = 0; img < nimg; ++img) {
        const unsigned img_off = (img & 1) * IMG;
#pragma unroll
        for (int c = 0; c < KW; ++c) {
            const int step = img * KW + c;
            const bool more_a = step + 2 < nsteps;
            const bool more_b = c == 0 && img + 1 < nimg;
            if (more_a) issue_a(abuf2);              // A first: the image issued after it may stay in flight
            if (more_b) issue_b((img + 1) & 1);
            const int tp = tap + c;
            unsigned keepv[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) keepv[j] = (mask[j] >> tp) & 1u;
            mfma_step<TM, TN, true>(acc, a_frag + abuf * A_STAGE, b_frag[c] + img_off, keepv);
            // A(step+1) -- and before an image's first step the image -- must have landed, for every wave
            if (c < KW - 1 && more_a && img + 1 < nimg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW + NIW) : "memory");
            else if (c == KW - 1 && more_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            abuf = abuf == 2 ? 0 : abuf + 1;
            abuf2 = abuf2 == 2 ? 0 : abuf2 + 1;
        }
        if (prm.cc_outer) {
            tap += KW;
            if (tap == prm.kT * prm.kH * KW) tap = 0;           // next chunk: the groups start over
        } else {
            tap += KW;
            if (img % prm.nchunk != prm.nchunk - 1) tap -= KW;  // same (kt,kh) group, next chunk
        }
    }
    epilogue<TM, TN, BM, BN>(prm, acc, (const float*)(lds + SHIFT_AT), R, Y, m0, n0, tm, wm, wn, tid, 0, 0, (float*)lds, tn, WGN);
#endif
}

// The same convolution with ONE image per (kt, chunk) for all nine (kh, kw) taps (round 4).  The images of kh = 0, 1, 2 are the
// same rows shifted by W: for a 256-column tile they overlap in all but 2 W rows, and the kernels on these layers run at the
// rate L2 -> LDS delivers (~4 TB/s over the chip for the forward AND for the 64-row input gradient, whose steps are three
// times as many and a third as long -- making its steps fatter changed nothing).  Rows n0 + (kt-pT) HW - W - 1 ... + BN + 2 W + 1
// are staged once (NI_P pieces: 24 for W <= 63, 20 for W <= 31); the fragment of tap (kh, kw) is read at row offset kh W + kw
// (the slot swizzle is conflict-free at any start row).  Image bytes per tile: 370 instead of 3 x 258 rows for W = 56.
template <int TM, int TN, int WGM, int WGN, int NI_P>
__global__ __launch_bounds__(256, 2) void conv_bf16_same9_kernel(Bf16Params prm, const __bf16* __restrict__ X,
                                                                const __bf16* __restrict__ Wp,
                                                                const float* __restrict__ shift,
                                                                const __bf16* __restrict__ R, __bf16* __restrict__ Y) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(WGM == 1 || (TM % 2) == 0, "row-block pairs must not straddle waves");
    constexpr int KW = 3;
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int NA_P = BM / 16;                                     // 1-KiB pieces of an A stage (NI_P: of the image)
    constexpr int KHW = 9;
    constexpr int NAW = (NA_P + 3) / 4, NIW = (NI_P + 3) / 4;
    constexpr int A_STAGE = BM * 64, IMG = NI_P * 1024;
    constexpr int IMG_AT = 3 * A_STAGE, SHIFT_AT = IMG_AT + 2 * IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int tile = xcd_tile(gridDim.x, blockIdx.x);
    const int tm = tile % prm.tiles_m, tn = tile / prm.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const int srcslot = ((lane & 3) ^ swz(lane >> 2)) * 8;
    int a_off[NAW], a_dst[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        int pa = wave + 4 * k;
        if (pa >= NA_P) pa -= 4;
        a_off[k] = (pa * 16 + (lane >> 2)) * 32 + srcslot;
        a_dst[k] = pa * 1024;
    }
    int i_row[NIW], i_dst[NIW];                      // image row this lane fills, per piece
#pragma unroll
    for (int k = 0; k < NIW; ++k) {
        int pi = wave + 4 * k;
        if (pi >= NI_P) pi -= 4;
        const int r = pi * 16 + (lane >> 2);
        i_row[k] = r < BN + 2 * prm.Wo + 2 ? n0 + r : -(1 << 30);    // rows past the last tap's last column: not fetched
        i_dst[k] = pi * 1024;
    }
    // border masks of this lane's 4 columns: bit tap = the tap reads inside the input.  One decode,
    // then +16 voxels per column block (carry into h, t).
    unsigned mask[TN];
    {
        const int p0 = n0 + wn * TN * 16 + (lane & 15);
        int rem = p0 % prm.ToHoWo;
        int t = rem / prm.HoWo;
        rem -= t * prm.HoWo;
        int h = rem / prm.Wo, w = rem - h * prm.Wo;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            unsigned m = 0;
            if (p0 + 16 * j < prm.P) {
                unsigned mt = 0, mh = 0, mw = 0;
                for (int a = 0; a < prm.kT; ++a) mt |= (unsigned)((unsigned)(t + a - prm.pT) < (unsigned)prm.Ti) << a;
                for (int b = 0; b < prm.kH; ++b) mh |= (unsigned)((unsigned)(h + b - prm.pH) < (unsigned)prm.Hi) << b;
                for (int c = 0; c < KW; ++c) mw |= (unsigned)((unsigned)(w + c - prm.pW) < (unsigned)prm.Wi) << c;
                int tap = 0;
                for (int a = 0; a < prm.kT; ++a)
                    for (int b = 0; b < prm.kH; ++b, tap += KW)
                        if (((mt >> a) & (mh >> b)) & 1u) m |= mw << tap;
            }
            mask[j] = m;
            w += 16;
            while (w >= prm.Wo) {
                w -= prm.Wo;
                if (++h == prm.Hi) { h = 0; if (++t == prm.Ti) t = 0; }
            }
        }
    }
    const __bf16* zero = (const __bf16*)zsv_zero_line;
    const size_t wq_step = (size_t)prm.Mp * 32;
    const __bf16* w_tile = Wp + (size_t)m0 * 32;

    // step = (kt, chunk cc, kh, kw); image = (kt, cc): rows n0 + (kt-pT)*HW - W - 1 ... + BN + 2W + 1 serve all nine taps
    const int nimg = prm.kT * prm.nchunk;
    const int nsteps = nimg * KHW;
    int a_kt = 0, a_cc = 0, a_t9 = 0;                // walk of the A issue
    auto issue_a = [&](int buf) {
        const __bf16* wq = w_tile + (size_t)((a_kt * KHW + a_t9) * prm.nchunk + a_cc) * wq_step;
        unsigned char* base = lds + buf * A_STAGE;
#pragma unroll
        for (int k = 0; k < NAW; ++k) __builtin_amdgcn_global_load_lds(wq + a_off[k], (lds_ptr_t)(base + a_dst[k]), 16, 0, 0);
        if (++a_t9 == KHW) {
            a_t9 = 0;
            if (prm.cc_outer) {
                if (++a_kt == prm.kT) { a_kt = 0; ++a_cc; }
            } else if (++a_cc == prm.nchunk) { a_cc = 0; ++a_kt; }
        }
    };
    int b_kt = 0, b_cc = 0;                          // walk of the B image issue
    auto issue_b = [&](int buf) {
        const int shift_rows = (b_kt - prm.pT) * prm.HoWo - prm.pH * prm.Wo - prm.pW;
        unsigned char* base = lds + IMG_AT + buf * IMG;
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            const int src = i_row[k] + shift_rows;
            const __bf16* ptr = (unsigned)src < (unsigned)prm.P ? X + ((size_t)src * prm.sW + b_cc * 32 + srcslot) : zero;
            __builtin_amdgcn_global_load_lds(ptr, (lds_ptr_t)(base + i_dst[k]), 16, 0, 0);
        }
        if (prm.cc_outer) {
            if (++b_kt == prm.kT) { b_kt = 0; ++b_cc; }
        } else if (++b_cc == prm.nchunk) { b_cc = 0; ++b_kt; }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned a_frag = lds_base + wm * TM * 16 * 64 + (lane & 15) * 64 + (((lane >> 4) ^ swz(lane & 15)) << 4);
    unsigned b_frag[KHW];                            // fragment base per (kh, kw): rows (l&15) + kh*W + kw of the column block
#pragma unroll
    for (int c = 0; c < KHW; ++c) {
        const int x = (lane & 15) + (c / KW) * prm.Wo + c % KW;
        b_frag[c] = lds_base + IMG_AT + (wn * TN * 16 + x) * 64 + (((lane >> 4) ^ swz(x)) << 4);
    }

    if (wave == 0) {
        const int l4 = lane < BM / 4 ? lane : BM / 4 - 1;
        __builtin_amdgcn_global_load_lds(shift + m0 + 4 * l4, (lds_ptr_t)(lds + SHIFT_AT), 16, 0, 0);
    }
    issue_b(0);
    issue_a(0);
    issue_a(1);                                      // nsteps >= 3
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW) : "memory");
    __builtin_amdgcn_s_barrier();
    int abuf = 0, abuf2 = 2;
    int kt = 0, cc = 0;
    for (int img = 0; img < nimg; ++img) {
        const unsigned img_off = (img & 1) * IMG;
        const int tap = kt * KHW;
#pragma unroll
        for (int c = 0; c < KHW; ++c) {
            const int step = img * KHW + c;
            const bool more_a = step + 2 < nsteps;
            const bool more_b = c == KHW - 3 && img + 1 < nimg;  // three steps ahead, as the per-(kt,kh) images were
            if (more_a) issue_a(abuf2);              // A first: the image issued after it may stay in flight
            if (more_b) issue_b((img + 1) & 1);
            const int tp = tap + c;
            unsigned keepv[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) keepv[j] = (mask[j] >> tp) & 1u;
            mfma_step<TM, TN, true>(acc, a_frag + abuf * A_STAGE, b_frag[c] + img_off, keepv);
            // A(step+1) -- and before an image's first step the image -- must have landed, for every wave.  In issue order
            // the next image sits between A(step+1) and A(step+2) for two steps: it may stay in flight there.
            if ((c == KHW - 3 || c == KHW - 2) && img + 1 < nimg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW + NIW) : "memory");
            else if (more_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            abuf = abuf == 2 ? 0 : abuf + 1;
            abuf2 = abuf2 == 2 ? 0 : abuf2 + 1;
        }
        if (prm.cc_outer) {
            if (++kt == prm.kT) { kt = 0; ++cc; }
        } else if (++cc == prm.nchunk) { cc = 0; ++kt; }
    }
    epilogue<TM, TN, BM, BN>(prm, acc, (const float*)(lds + SHIFT_AT), R, Y, m0, n0, tm, wm, wn, tid, 0, 0, (float*)lds, tn, WGN);
#endif
}

// Temporal 3x1x1 stride-1 convolutions (Conv2Plus1D's second half, resnet.py:50-52): the workgroup tile
// is TT output frames x HB (h,w) positions (TT * HB = 256 columns), so ONE LDS image of TT+2 input
// frames x HB positions serves the three taps -- a tap is a shift of HB rows, always 16-row aligned --
// and the input crosses L2 -> LDS 1.25x (TT = 8) instead of 3x.  A frame outside the clip is a zero
// image row block; the test is uniform per 16-column block.  Steps = (chunk, kt): A ring of 3 per step,
// image ring of 2 per chunk, as in conv_bf16_same_kernel.
template <int TM, int TN, int WGM, int WGN, int TT>
__global__ __launch_bounds__(256, 2) void conv_bf16_tsame_kernel(Bf16Params prm, const __bf16* __restrict__ X,
                                                                 const __bf16* __restrict__ Wp,
                                                                 const float* __restrict__ shift,
                                                                 const __bf16* __restrict__ R, __bf16* __restrict__ Y) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WGM == 1 && WGN == 4 && TN == 4, "256 columns = 4 waves x 4 blocks");
    constexpr int KT = 3;
    constexpr int BM = 16 * TM, BN = 256, HB = BN / TT;
    constexpr int HB_SHIFT = HB == 32 ? 5 : 4;
    static_assert(HB == 32 || HB == 16, "positions per frame row block");
    constexpr int NA_P = BM / 16, NI_P = (TT + 2) * HB / 16;
    constexpr int NAW = (NA_P + 3) / 4, NIW = (NI_P + 3) / 4;
    constexpr int A_STAGE = BM * 64, IMG = NI_P * 1024;
    constexpr int IMG_AT = 3 * A_STAGE, SHIFT_AT = IMG_AT + 2 * IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_tile(gridDim.x, blockIdx.x);
    const int tm = tile % prm.tiles_m, tn = tile / prm.tiles_m;
    const int m0 = tm * BM;
    // column tile -> (clip, frame block, position block); positions fastest
    const int T = prm.Ti, HW = prm.HoWo;
    const int hbs = HW / HB, tbs = T / TT;
    const int hb = tn % hbs, tb = (tn / hbs) % tbs, n = tn / (hbs * tbs);
    const int t0 = tb * TT, hw0 = hb * HB;
    const int n0 = (n * T + t0) * HW + hw0;          // voxel of column 0

    const int srcslot = ((lane & 3) ^ swz(lane >> 2)) * 8;
    int a_off[NAW], a_dst[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        int pa = wave + 4 * k;
        if (pa >= NA_P) pa -= 4;
        a_off[k] = (pa * 16 + (lane >> 2)) * 32 + srcslot;
        a_dst[k] = pa * 1024;
    }
    long i_src[NIW];                                 // element offset of this lane's image row (chunk 0), or -1
    int i_dst[NIW];
#pragma unroll
    for (int k = 0; k < NIW; ++k) {
        int pi = wave + 4 * k;
        if (pi >= NI_P) pi -= 4;
        const int r = pi * 16 + (lane >> 2);
        const int tin = t0 - 1 + (r >> HB_SHIFT);
        i_src[k] = (unsigned)tin < (unsigned)T ? ((long)(n * T + tin) * HW + hw0 + (r & (HB - 1))) * prm.sW + srcslot : -1;
        i_dst[k] = pi * 1024;
    }
    // validity of tap kt for this wave's column block j: output frame t0 + ((wave*4+j)*16 >> HB_SHIFT)
    unsigned keep[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int tf = t0 + (((wave * TN + j) * 16) >> HB_SHIFT);
        unsigned m = 0;
        for (int a = 0; a < KT; ++a) m |= (unsigned)((unsigned)(tf + a - 1) < (unsigned)T) << a;
        keep[j] = m;
    }
    const __bf16* zero = (const __bf16*)zsv_zero_line;
    const size_t wq_step = (size_t)prm.Mp * 32;
    const __bf16* w_tile = Wp + (size_t)m0 * 32;

    const int nimg = prm.nchunk, nsteps = nimg * KT;
    int a_cc = 0, a_kt = 0;
    auto issue_a = [&](int buf) {
        const __bf16* wq = w_tile + (size_t)(a_kt * prm.nchunk + a_cc) * wq_step;
        unsigned char* base = lds + buf * A_STAGE;
#pragma unroll
        for (int k = 0; k < NAW; ++k) __builtin_amdgcn_global_load_lds(wq + a_off[k], (lds_ptr_t)(base + a_dst[k]), 16, 0, 0);
        if (++a_kt == KT) { a_kt = 0; ++a_cc; }
    };
    int b_cc = 0;
    auto issue_b = [&](int buf) {
        unsigned char* base = lds + IMG_AT + buf * IMG;
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            const __bf16* ptr = i_src[k] >= 0 ? X + (i_src[k] + b_cc * 32) : zero;
            __builtin_amdgcn_global_load_lds(ptr, (lds_ptr_t)(base + i_dst[k]), 16, 0, 0);
        }
        ++b_cc;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned frag = (lane & 15) * 64 + (((lane >> 4) ^ swz(lane & 15)) << 4);
    const unsigned a_frag = lds_base + frag;
    const unsigned b_frag = lds_base + IMG_AT + wave * TN * 1024 + frag;      // + kt * HB rows per tap

    if (wave == 0) {
        const int l4 = lane < BM / 4 ? lane : BM / 4 - 1;
        __builtin_amdgcn_global_load_lds(shift + m0 + 4 * l4, (lds_ptr_t)(lds + SHIFT_AT), 16, 0, 0);
    }
    issue_b(0);
    issue_a(0);
    issue_a(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW) : "memory");
    __builtin_amdgcn_s_barrier();
    int abuf = 0, abuf2 = 2;
    for (int img = 0; img < nimg; ++img) {
        const unsigned img_off = (img & 1) * IMG;
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const int step = img * KT + c;
            const bool more_a = step + 2 < nsteps;
            const bool more_b = c == 0 && img + 1 < nimg;
            if (more_a) issue_a(abuf2);
            if (more_b) issue_b((img + 1) & 1);
            unsigned keepv[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) keepv[j] = (keep[j] >> c) & 1u;
            mfma_step<TM, TN, true>(acc, a_frag + abuf * A_STAGE, b_frag + img_off + c * HB * 64, keepv);
            if (c < KT - 1 && more_a && img + 1 < nimg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW + NIW) : "memory");
            else if (c == KT - 1 && more_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            abuf = abuf == 2 ? 0 : abuf + 1;
            abuf2 = abuf2 == 2 ? 0 : abuf2 + 1;
        }
    }
    epilogue<TM, TN, BM, BN>(prm, acc, (const float*)(lds + SHIFT_AT), R, Y, m0, n0, tm, 0, wave, tid, HB_SHIFT, HW, (float*)lds, tn, 4);
#endif
}

// Wp[q][Mp][32] <- w[Cout][Cin][kT][kH][kW] * scale[cout]; then Mp fp32 shifts.
__global__ void pack_bf16_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                 const float* __restrict__ shift, __bf16* __restrict__ wp, float* __restrict__ shift_out,
                                 int M, int Mp, int bm, int Cin, int taps, int nchunk, int kW, int folded, long total,
                                 int transposed = 0) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < Mp) shift_out[idx] = (idx < M && shift != nullptr) ? shift[idx] : 0.f;     // channel order
    if (idx >= total) return;
    const int k = (int)(idx & 31);
    const long rq = idx >> 5;
    const int prow = (int)(rq % Mp);               // packed row -> the channel it holds
    const int row = prow / bm * bm + tile_channel(prow % bm, bm);
    const int q = (int)(rq / Mp);
    float v = 0.f;
    if (row < M) {
        const float s = scale != nullptr ? scale[row] : 1.f;
        if (folded) {                       // q = kt*kH + kh, k = 4*kw + c
            const int kwi = k >> 2, c = k & 3;
            if (kwi < kW && c < Cin) v = w[(((long)row * Cin + c) * taps + q) * kW + kwi] * s;
        } else {
            const int tap = q / nchunk, ci = (q - tap * nchunk) * 32 + k;
            // transposed: `w` is the FORWARD weight (Cin, M, taps) of the convolution whose input gradient this problem is:
            // channel roles swapped, taps reversed (flip along kt, kh and kw = the linear tap index read backwards)
            if (ci < Cin) v = (transposed ? w[((long)ci * M + row) * taps + (taps - 1 - tap)] : w[((long)row * Cin + ci) * taps + tap]) * s;
        }
    }
    wp[idx] = (__bf16)v;
}

// (N,3,T,H,W) fp32 -> [N][T][Hp][Wp][4] bf16 with a zero border of (padH, padW) (and zero 4th channel)
__global__ void clip_to_bf16_kernel(const float* __restrict__ x, int C, int T, int H, int W, int padH, int padW, int Hp,
                                    int Wpix, long total, __bf16* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int wp = (int)(idx % Wpix);
    long r = idx / Wpix;
    const int hp = (int)(r % Hp);
    r /= Hp;
    const int t = (int)(r % T);
    const long n = r / T;
    const int h = hp - padH, wq = wp - padW;
    bf16x4 o;
    o[0] = o[1] = o[2] = o[3] = (__bf16)0.f;
    if ((unsigned)h < (unsigned)H && (unsigned)wq < (unsigned)W) {
        const long S = (long)T * H * W;
        const float* src = x + n * C * S + ((long)t * H + h) * W + wq;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) o[c] = (__bf16)src[c * S];
    }
    *(bf16x4*)(out + idx * 4) = o;
}

// [N][S][Cp] bf16 -> (N, C) fp32 mean over S (resnet.py:251 AdaptiveAvgPool3d(1))
__global__ __launch_bounds__(256) void meanpool_bf16_kernel(const __bf16* __restrict__ x, int S, int Cp, int C,
                                                            float* __restrict__ out) {
    __shared__ float part[4][64];
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int s = slice; s < S; s += 4) acc += (float)x[((size_t)n * S + s) * Cp + c];
    part[slice][threadIdx.x & 63] = acc;
    __syncthreads();
    if (slice == 0 && c < C)
        out[(size_t)n * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)S;
}

// nn.MaxPool3d with kernel == stride on channels-last bf16 (network.py:148-163 between the C3D convolutions): one thread per
// (output voxel, 8-channel octet), 16-byte loads; taps outside the input are skipped (-inf padding).  bf16 -> float is exact and
// monotone, so the maximum of the floats rounds back to one of the inputs.
__global__ void maxpool3d_bf16_kernel(const u32x4* __restrict__ x, int Ti, int Hi, int Wi, int G, int kT, int kH, int kW, int pT, int pH,
                                      int pW, int To, int Ho, int Wo, long total, u32x4* __restrict__ y) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int o = (int)(idx % G);
    long r = idx / G;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int to = (int)(r % To);
    const long n = r / To;
    float best[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = -INFINITY;
    for (int a = 0; a < kT; ++a) {
        const int t = to * kT + a - pT;
        if ((unsigned)t >= (unsigned)Ti) continue;
        for (int b = 0; b < kH; ++b) {
            const int h = ho * kH + b - pH;
            if ((unsigned)h >= (unsigned)Hi) continue;
            for (int c = 0; c < kW; ++c) {
                const int w = wo * kW + c - pW;
                if ((unsigned)w >= (unsigned)Wi) continue;
                const u32x4 v = x[(((n * Ti + t) * Hi + h) * (long)Wi + w) * G + o];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    best[2 * q] = fmaxf(best[2 * q], __builtin_bit_cast(float, v[q] << 16));
                    best[2 * q + 1] = fmaxf(best[2 * q + 1], __builtin_bit_cast(float, v[q] & 0xffff0000u));
                }
            }
        }
    }
    u32x4 out;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        out[q] = (__builtin_bit_cast(unsigned, best[2 * q]) >> 16) | (__builtin_bit_cast(unsigned, best[2 * q + 1]) & 0xffff0000u);
    y[idx] = out;
}

// ---- host side -----------------------------------------------------------------------------------
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline bool bf16_folded(const zsv_conv_desc* d) { return d->Cin <= 4; }
static inline int bf16_cin_pitch(const zsv_conv_desc* d) { return bf16_folded(d) ? 4 : round_up(d->Cin, 32); }

// row tile height: a function of Cout only, so that weights are packed once per layer
static int bf16_bm(int M) {
    if (M <= 64) return 64;
    const int p128 = round_up(M, 128), p144 = round_up(M, 144);
    return p144 < p128 ? 144 : 128;
}

static int bf16_check(const zsv_conv_desc* d) {
    if (d == nullptr) return ZSV_E_NULL;
    if (bf16_folded(d)) {
        // the border is materialised in the input (pH = pW = 0) and may be wider than the taps need: a
        // chunk reads 8 pixels starting at wo*sW, so Wi >= (Wo-1)*sW + 8 while Wo follows the true frame
        if (d->pH != 0 || d->pW != 0 || d->kW > 8) return ZSV_E_UNSUPPORTED;
        if (d->Wo <= 0 || d->sW <= 0 || (d->Wo - 1) * d->sW + 8 > d->Wi) return ZSV_E_BAD_SHAPE;
        zsv_conv_desc tight = *d;
        tight.Wi = (d->Wo - 1) * d->sW + d->kW;
        const int st = conv_check(&tight);
        if (st != ZSV_OK) return st;
    } else {
        const int st = conv_check(d);
        if (st != ZSV_OK) return st;
    }
    if ((bf16_folded(d) ? d->kT * d->kH : d->kT * d->kH * d->kW) > 32) return ZSV_E_UNSUPPORTED;   // tap validity mask
    const long in_elems = (long)d->N * d->Ti * d->Hi * d->Wi * bf16_cin_pitch(d);
    const long out_vox = (long)d->N * d->To * d->Ho * d->Wo;
    if (in_elems >= (1L << 31) || out_vox * round_up(d->Cout, 32) >= (1L << 31)) return ZSV_E_TOO_LARGE;
    return ZSV_OK;
}

template <int TM, int TN, int WGM, int WGN>
static int bf16_launch(Bf16Params& p, hipStream_t stream, const __bf16* x, const __bf16* wp, const float* shift,
                       const __bf16* r, __bf16* y) {
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int LDS_BYTES = 3 * (BM + BN) * 64 + 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_bf16_kernel<TM, TN, WGM, WGN>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    p.tiles_m = p.Mp / BM;
    p.tiles_n = (p.P + BN - 1) / BN;
    hipLaunchKernelGGL((conv_bf16_kernel<TM, TN, WGM, WGN>), dim3(p.tiles_m * p.tiles_n), dim3(256), LDS_BYTES, stream, p, x,
                       wp, shift, r, y);
    return launch_status();
}

template <int TM, int TN, int WGM, int WGN>
static int bf16_same_launch(Bf16Params& p, hipStream_t stream, const __bf16* x, const __bf16* wp, const float* shift,
                            const __bf16* r, __bf16* y) {
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int LDS_BYTES = 3 * BM * 64 + 2 * ((BN + 2 + 15) / 16) * 1024 + 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_bf16_same_kernel<TM, TN, WGM, WGN>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    p.tiles_m = p.Mp / BM;
    p.tiles_n = (p.P + BN - 1) / BN;
    hipLaunchKernelGGL((conv_bf16_same_kernel<TM, TN, WGM, WGN>), dim3(p.tiles_m * p.tiles_n), dim3(256), LDS_BYTES, stream,
                       p, x, wp, shift, r, y);
    return launch_status();
}

template <int TM, int TN, int WGM, int WGN, int NI_P>
static int bf16_same9_launch_n(Bf16Params& p, hipStream_t stream, const __bf16* x, const __bf16* wp, const float* shift,
                               const __bf16* r, __bf16* y) {
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int LDS_BYTES = 3 * BM * 64 + 2 * NI_P * 1024 + 1024;
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_bf16_same9_kernel<TM, TN, WGM, WGN, NI_P>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    p.tiles_m = p.Mp / BM;
    p.tiles_n = (p.P + BN - 1) / BN;
    hipLaunchKernelGGL((conv_bf16_same9_kernel<TM, TN, WGM, WGN, NI_P>), dim3(p.tiles_m * p.tiles_n), dim3(256), LDS_BYTES, stream,
                       p, x, wp, shift, r, y);
    return launch_status();
}

// one image for the nine (kh, kw) taps: kH = 3 with pH = 1 and a row short enough for the image to fit next to the A ring
// (and at least four K chunks: with two -- the 64-channel forward of layer1 -- a tile has two images and starts with the bigger
// one: measured 0.341 against 0.305 ms there)
// and ONE row tile or 3x3x3 taps: with several row tiles of 1x3x3 taps (the 230 ... 576-channel forwards of the evaluation engine) the
// 32-frame forward was 1 % slower with it (6.31 against 6.25 ms), C3D's 7 % faster (3.10 against 3.32 ms) -- measured, not modelled
static bool bf16_same9_applicable(const zsv_conv_desc* d, int row_tiles) {
    const char* e = ZSV_KNOB(BF16_SAME9_MIN_CHUNKS);
    const int min_chunks = e ? atoi(e) : 4;
    return d->kH == 3 && d->pH == 1 && d->Wi <= 63 && (d->Cin + 31) / 32 >= min_chunks && (row_tiles == 1 || d->kT == 3) &&
           ZSV_KNOB(BF16_NO_SAME9) == nullptr;
}

template <int TM>
static int bf16_same9_launch(const zsv_conv_desc* d, Bf16Params& p, hipStream_t stream, const __bf16* x, const __bf16* wp,
                             const float* shift, const __bf16* r, __bf16* y) {
    return d->Wi <= 31 ? bf16_same9_launch_n<TM, 4, 1, 4, 20>(p, stream, x, wp, shift, r, y)
                       : bf16_same9_launch_n<TM, 4, 1, 4, 24>(p, stream, x, wp, shift, r, y);
}

// stride 1, output extents = input extents, kW = 3 with pW = 1: taps are flattened shifts
static bool bf16_same_applicable(const zsv_conv_desc* d) {
    return !bf16_folded(d) && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->kW == 3 && d->pW == 1 && d->To == d->Ti &&
           d->Ho == d->Hi && d->Wo == d->Wi && ZSV_KNOB(BF16_NO_SAME) == nullptr;
}

template <int TM, int TT>
static int bf16_tsame_launch(Bf16Params& p, const zsv_conv_desc* d, hipStream_t stream, const __bf16* x, const __bf16* wp,
                             const float* shift, const __bf16* r, __bf16* y) {
    constexpr int BM = 16 * TM, HB = 256 / TT;
    constexpr int LDS_BYTES = 3 * BM * 64 + 2 * ((TT + 2) * HB / 16) * 1024 + 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_bf16_tsame_kernel<TM, 4, 1, 4, TT>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    p.tiles_m = p.Mp / BM;
    p.tiles_n = d->N * (d->Ti / TT) * (d->Hi * d->Wi / HB);
    hipLaunchKernelGGL((conv_bf16_tsame_kernel<TM, 4, 1, 4, TT>), dim3(p.tiles_m * p.tiles_n), dim3(256), LDS_BYTES, stream, p,
                       x, wp, shift, r, y);
    return launch_status();
}

// 3x1x1, stride 1, pad (1,0,0): frames-x-positions tiles if the clip divides into them; returns TT or 0
static int bf16_tsame_frames(const zsv_conv_desc* d) {
    if (bf16_folded(d) || d->kT != 3 || d->kH != 1 || d->kW != 1 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT != 1 ||
        d->pH != 0 || d->pW != 0 || ZSV_KNOB(BF16_NO_TSAME))
        return 0;
    const int HW = d->Hi * d->Wi;
    if (d->Ti % 8 == 0 && HW % 32 == 0) return 8;
    if (d->Ti % 16 == 0 && HW % 16 == 0) return 16;
    return 0;
}

}  // namespace zsv

using namespace zsv;

extern "C" {

int32_t zsv_bf16_channel_pitch(int32_t channels) { return channels <= 4 ? 4 : round_up(channels, 32); }

size_t zsv_conv3d_bf16_blob_bytes(const zsv_conv_desc* d) {
    if (d == nullptr || bf16_check(d) != ZSV_OK) return 0;
    const int bm = bf16_bm(d->Cout), Mp = round_up(d->Cout, bm);
    const int taps = bf16_folded(d) ? d->kT * d->kH : d->kT * d->kH * d->kW;
    const int nchunk = bf16_folded(d) ? 1 : round_up(d->Cin, 32) / 32;
    return (size_t)taps * nchunk * Mp * 32 * 2 + (size_t)Mp * 4;
}

int zsv_conv3d_bf16_pack(const zsv_conv_desc* d, const float* w, const float* scale, const float* shift, void* blob,
                         void* stream) {
    if (d == nullptr) return ZSV_E_NULL;
    const int st = bf16_check(d);
    if (st != ZSV_OK) return st;
    if (w == nullptr || blob == nullptr) return ZSV_E_NULL;
    const bool folded = bf16_folded(d);
    const int bm = bf16_bm(d->Cout), Mp = round_up(d->Cout, bm);
    const int taps = folded ? d->kT * d->kH : d->kT * d->kH * d->kW;
    const int nchunk = folded ? 1 : round_up(d->Cin, 32) / 32;
    const long total = (long)taps * nchunk * Mp * 32;
    __bf16* wp = (__bf16*)blob;
    float* shift_out = (float*)((char*)blob + total * 2);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, scale,
                       shift, wp, shift_out, d->Cout, Mp, bm, d->Cin, taps, nchunk, d->kW, folded ? 1 : 0, total);
    return launch_status();
}

int zsv_conv3d_bf16_pack_dgrad(const zsv_conv_desc* d, const float* w_fwd, void* blob, void* stream) {
    if (d == nullptr) return ZSV_E_NULL;
    const int st = bf16_check(d);
    if (st != ZSV_OK) return st;
    if (w_fwd == nullptr || blob == nullptr) return ZSV_E_NULL;
    if (bf16_folded(d)) return ZSV_E_UNSUPPORTED;
    const int bm = bf16_bm(d->Cout), Mp = round_up(d->Cout, bm);
    const int taps = d->kT * d->kH * d->kW;
    const int nchunk = round_up(d->Cin, 32) / 32;
    const long total = (long)taps * nchunk * Mp * 32;
    __bf16* wp = (__bf16*)blob;
    float* shift_out = (float*)((char*)blob + total * 2);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_fwd, nullptr,
                       nullptr, wp, shift_out, d->Cout, Mp, bm, d->Cin, taps, nchunk, d->kW, 0, total, 1);
    return launch_status();
}

static int bf16_fwd_impl(const zsv_conv_desc* d, const void* x, const void* blob, const void* residual, int fuse_relu, void* y,
                         float* stat, int32_t stat_rows_capacity, int32_t* stat_rows, void* stream);

int zsv_conv3d_bf16_fwd(const zsv_conv_desc* d, const void* x, const void* blob, const void* residual, int fuse_relu,
                        void* y, void* stream) {
    return bf16_fwd_impl(d, x, blob, residual, fuse_relu, y, nullptr, 0, nullptr, stream);
}

// rows of the statistics array a forward of this geometry may write (an upper bound: the smallest column tile is 128 voxels)
int32_t zsv_conv3d_bf16_stat_rows(const zsv_conv_desc* d) {
    if (d == nullptr || bf16_check(d) != ZSV_OK) return 0;
    const long P = (long)d->N * d->To * d->Ho * d->Wo;
    return (int32_t)((P + 127) / 128);
}

int zsv_conv3d_bf16_fwd_stats(const zsv_conv_desc* d, const void* x, const void* blob, void* y, float* bn_partials,
                              int32_t rows_capacity, int32_t* rows, void* stream) {
    if (bn_partials == nullptr || rows == nullptr) return ZSV_E_NULL;
    return bf16_fwd_impl(d, x, blob, nullptr, 0, y, bn_partials, rows_capacity, rows, stream);
}

static int bf16_fwd_impl(const zsv_conv_desc* d, const void* x, const void* blob, const void* residual, int fuse_relu, void* y,
                         float* stat, int32_t stat_rows_capacity, int32_t* stat_rows, void* stream) {
    if (d == nullptr) return ZSV_E_NULL;
    const int st = bf16_check(d);
    if (st != ZSV_OK) return st;
    if (x == nullptr || blob == nullptr || y == nullptr) return ZSV_E_NULL;
    const bool folded = bf16_folded(d);
    Bf16Params p;
    const int bm = bf16_bm(d->Cout);
    p.M = d->Cout;
    p.Mp = round_up(d->Cout, bm);
    p.CoutP = round_up(d->Cout, 32);
    const int cp = bf16_cin_pitch(d);
    p.nchunk = folded ? 1 : cp / 32;
    p.kT = d->kT; p.kH = d->kH; p.kW = folded ? 1 : d->kW;
    p.nq = p.kT * p.kH * p.kW * p.nchunk;
    p.sW = cp; p.sH = d->Wi * cp; p.sT = d->Hi * d->Wi * cp;
    p.sN = (long)d->Ti * d->Hi * d->Wi * cp;
    p.Ti = d->Ti; p.Hi = d->Hi; p.Wi = d->Wi;
    p.Wo = d->Wo; p.HoWo = d->Ho * d->Wo; p.ToHoWo = d->To * d->Ho * d->Wo;
    p.strT = d->sT; p.strH = d->sH; p.strW = d->sW;
    p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
    p.P = d->N * p.ToHoWo;
    p.relu = fuse_relu ? 1 : 0;
    p.stat = stat;
    if (stat != nullptr) {
        // one row per column tile of the kernel chosen below (256 voxels, 128 in the small-problem form)
        if (residual != nullptr || fuse_relu || stat_rows_capacity < zsv_conv3d_bf16_stat_rows(d)) return ZSV_E_WORKSPACE;
        *stat_rows = (int32_t)(((long)(p.Mp / 128) * ((p.P + 255) / 256) < 384 && bm == 128) ? (p.P + 127) / 128 : (p.P + 255) / 256);
    }
    // (with >= 3 chunks of 32 input channels the image rows of all chunks of a (kt,kh) group no longer fit the XCD's L2 next to the
    // other workgroups': 1.19 GB fetched for 353 MB of dz on layer1's input gradient, profiles/r04_bf16_training_kernels_pmc.json)
    p.cc_outer = (p.nchunk >= 3 && ZSV_KNOB(BF16_GROUP_OUTER) == nullptr) ? 1 : 0;
    const __bf16* wp = (const __bf16*)blob;
    const float* shift = (const float*)((const char*)blob + (size_t)p.nq * p.Mp * 32 * 2);
    const __bf16* xb = (const __bf16*)x;
    const __bf16* rb = (const __bf16*)residual;
    __bf16* yb = (__bf16*)y;
    hipStream_t s = (hipStream_t)stream;
    const bool small = (long)(p.Mp / 128) * ((p.P + 255) / 256) < 384;      // too few 128x256 tiles to fill the chip
    if (const int tt = bf16_tsame_frames(d); tt != 0 && (bm == 64 || (bm == 128 && !small))) {
        if (bm == 64) return tt == 8 ? bf16_tsame_launch<4, 8>(p, d, s, xb, wp, shift, rb, yb)
                                     : bf16_tsame_launch<4, 16>(p, d, s, xb, wp, shift, rb, yb);
        // (a 144-row form of this kernel for the input gradient of layer1's temporal convolutions -- 64 -> 144 channels in the bf16
        // training step -- was built and measured: 265 us against 215 us for the per-tap kernel on that shape; not kept)
        return tt == 8 ? bf16_tsame_launch<8, 8>(p, d, s, xb, wp, shift, rb, yb)
                       : bf16_tsame_launch<8, 16>(p, d, s, xb, wp, shift, rb, yb);
    }
    // (the 64-row wave tiles of the small-P configuration have too few MFMAs per step to hide the
    // fragment masking of the shared-image kernel: measured slower there)
    if (bf16_same_applicable(d) && bm != 64 && !(bm == 128 && small)) {
        if (bf16_same9_applicable(d, p.Mp / bm))
            return bm == 144 ? bf16_same9_launch<9>(d, p, s, xb, wp, shift, rb, yb) : bf16_same9_launch<8>(d, p, s, xb, wp, shift, rb, yb);
        if (bm == 144) return bf16_same_launch<9, 4, 1, 4>(p, s, xb, wp, shift, rb, yb);
        return bf16_same_launch<8, 4, 1, 4>(p, s, xb, wp, shift, rb, yb);
    }
    // 64 produced channels with many voxels: the input gradient of layer1's spatial convolutions in the bf16 training step
    // (144 -> 64 channels, K = 9 x 160; amp.py) -- the per-tap kernel gathers every input row nine times for 64 rows of MFMAs
    // (a 64 x 512 tile -- eight column blocks per wave, half the LDS-DMA bytes per MFMA -- was built for this case and measured
    // SLOWER: 0.538 vs 0.458 ms on layer1's input gradient, step 21.4 vs 21.1 ms on one device; the 80 KB of LDS per workgroup and
    // the 33-piece image fill cost more than the bytes saved.  mfma_step keeps its 8-block form.)
    if (bf16_same_applicable(d) && bm == 64 && p.P >= 256 * 512 && ZSV_KNOB(BF16_NO_SAME64) == nullptr)
        return bf16_same9_applicable(d, p.Mp / bm) ? bf16_same9_launch<4>(d, p, s, xb, wp, shift, rb, yb)
                                        : bf16_same_launch<4, 4, 1, 4>(p, s, xb, wp, shift, rb, yb);
    if (bm == 64) return bf16_launch<4, 4, 1, 4>(p, s, xb, wp, shift, rb, yb);
    if (bm == 144) return bf16_launch<9, 4, 1, 4>(p, s, xb, wp, shift, rb, yb);
    if (small) return bf16_launch<4, 4, 2, 2>(p, s, xb, wp, shift, rb, yb);
    return bf16_launch<8, 4, 1, 4>(p, s, xb, wp, shift, rb, yb);
}

int zsv_clip_to_bf16(const float* x, int32_t N, int32_t C, int32_t T, int32_t H, int32_t W, int32_t padH, int32_t padW,
                     int32_t Hp, int32_t Wp, void* out, void* stream) {
    if (N <= 0 || C <= 0 || C > 4 || T <= 0 || H <= 0 || W <= 0 || padH < 0 || padW < 0 || Hp < H + padH || Wp < W + padW)
        return ZSV_E_BAD_SHAPE;
    if (x == nullptr || out == nullptr) return ZSV_E_NULL;
    const long total = (long)N * T * Hp * Wp;
    if (total * 4 >= (1L << 31)) return ZSV_E_TOO_LARGE;
    hipLaunchKernelGGL(clip_to_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, C, T,
                       H, W, padH, padW, Hp, Wp, total, (__bf16*)out);
    return launch_status();
}

int zsv_maxpool3d_bf16(const void* x, int32_t N, int32_t C, int32_t Ti, int32_t Hi, int32_t Wi, int32_t kT, int32_t kH, int32_t kW,
                       int32_t pT, int32_t pH, int32_t pW, int32_t To, int32_t Ho, int32_t Wo, void* y, void* stream) {
    if (N <= 0 || C <= 4 || Ti <= 0 || Hi <= 0 || Wi <= 0 || kT <= 0 || kH <= 0 || kW <= 0 || pT < 0 || pH < 0 || pW < 0)
        return ZSV_E_BAD_SHAPE;
    if (2 * pT > kT || 2 * pH > kH || 2 * pW > kW) return ZSV_E_BAD_SHAPE;            // (every window holds at least one input voxel)
    if (To != (Ti + 2 * pT - kT) / kT + 1 || Ho != (Hi + 2 * pH - kH) / kH + 1 || Wo != (Wi + 2 * pW - kW) / kW + 1 || To <= 0 ||
        Ho <= 0 || Wo <= 0)
        return ZSV_E_BAD_SHAPE;
    if (x == nullptr || y == nullptr) return ZSV_E_NULL;
    const int G = round_up(C, 32) / 8;
    const long total = (long)N * To * Ho * Wo * G;
    if ((long)N * Ti * Hi * Wi * G >= (1L << 31)) return ZSV_E_TOO_LARGE;
    hipLaunchKernelGGL(maxpool3d_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x,
                       Ti, Hi, Wi, G, kT, kH, kW, pT, pH, pW, To, Ho, Wo, total, (u32x4*)y);
    return launch_status();
}

int zsv_meanpool_bf16(const void* x, int32_t N, int32_t S, int32_t C, float* out, void* stream) {
    if (N <= 0 || S <= 0 || C <= 4) return ZSV_E_BAD_SHAPE;
    if (x == nullptr || out == nullptr) return ZSV_E_NULL;
    hipLaunchKernelGGL(meanpool_bf16_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, S,
                       round_up(C, 32), C, out);
    return launch_status();
}

}  // extern "C"
