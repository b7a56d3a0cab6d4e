// conv_wgrad_dma.hip -- weight gradient of stride-1 "same" convolutions with LDS-DMA staging.
//
//   dW[co][ci][tap] = sum_p dY[co][p] * X[ci][p + d(tap)]      (aten::convolution_backward, weight part,
//                                                               for resnet.py:23-30,40-52,63-70)
// For stride 1 and equal input/output extents a tap is a constant shift d(tap) = (kt-pT)*H*W +
// (kh-pH)*W + (kw-pW) of the flattened voxel index inside one clip, plus a border test.  Both GEMM
// operands are then rows that are CONTIGUOUS ALONG THE REDUCTION (voxels): dY[co][.] and X[ci][. + d].
//
//   * MFMA k order: v_mfma_f32_16x16x4_f32 takes k = lane>>4 from each lane; any bijection of the 16
//     voxels of a chunk onto (step, lane>>4) is valid as long as A and B agree.  With k = 4*(lane>>4) +
//     step a lane needs 4 CONSECUTIVE voxels of its row for the 4 steps: one ds_read_b128 per fragment
//     and 16 voxels (instead of one ds_read_b32 per fragment and step).
//   * So the LDS image is [row][16 voxels] = 64-byte rows, written by global_load_lds_dwordx4 straight
//     from the tensors (16-byte slot XOR-swizzled on the source side, as in conv_bf16.hip): no staging
//     registers, no ds_write, no per-chunk address arithmetic beyond a pointer increment.
//   * The border test is applied to the B fragment: a per-voxel bit mask over the taps (built once per
//     call by wgrad_vmask_kernel, S words) is DMA'd next to the operands; a lane zeroes the voxels of
//     its 4 whose tap falls outside the input.  Rows read past a clip plane pick up the neighbouring
//     plane's values, which the same mask discards.
// Output: the same per-slice slabs OUT[slice][co][tap*Cpad + ci] as conv_wgrad.hip (slab_sum_kernel
// reduces them in slice order: bitwise reproducible).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "conv_params.h"
#include "zsv_common.h"
#include "zsv_hip.h"
#include "knobs.h"

namespace zsv {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ f32x4 zsv_wgrad_zero_line[8];      // 128 zero bytes: rows beyond Cout / Cin / the tensor

struct WgradDmaParams {
    int M, Cin, Cpad, taps, Kp;       // Kp = taps * Cpad columns, tap-major
    int S, HW, W;                     // voxels per clip / frame / row (input extents = output extents)
    int kH, kW, pT, pH, pW;
    int chunks_total, chunks_per_slice;      // 16-voxel chunks
    int walk_T;                               // > 0: chunks walk the T frames of one 16-voxel (h,w) segment first
    long x_elems;
    int tiles_m, tiles_mn;
};

__host__ __device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }

template <int OFF>
__device__ __forceinline__ void lds_read128f(f32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_read128u(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int N>
__device__ __forceinline__ void lds_wait1(f32x4& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

// tap-validity bits of every voxel of one clip: bit tap set <=> (t+kt-pT, h+kh-pH, w+kw-pW) is inside
__global__ void wgrad_vmask_kernel(int S, int T, int H, int W, int kT, int kH, int kW, int pT, int pH, int pW,
                                   unsigned* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S) return;
    const int t = p / (H * W), r = p - t * H * W, h = r / W, w = r - h * W;
    unsigned m = 0;
    int tap = 0;
    for (int a = 0; a < kT; ++a)
        for (int b = 0; b < kH; ++b)
            for (int c = 0; c < kW; ++c, ++tap)
                m |= (unsigned)((unsigned)(t + a - pT) < (unsigned)T && (unsigned)(h + b - pH) < (unsigned)H &&
                                (unsigned)(w + c - pW) < (unsigned)W) << tap;
    out[p] = m;
}

// KB = 16-voxel chunks per barrier (2 for the small tiles, whose 16-32 MFMAs per chunk would otherwise be
// shorter than the barrier + DMA turn-around)
template <int TM, int TN, int KB>
__global__ __launch_bounds__(256, (TM * TN > 27 ? 2 : 3)) void conv_wgrad_dma_kernel(WgradDmaParams prm, const float* __restrict__ X,
                                                                const float* __restrict__ DY,
                                                                const unsigned* __restrict__ VM,
                                                                float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(TM >= 4, "piece distribution");
    constexpr int BM = 16 * TM, BN = 64 * TN;        // 4 waves side by side along the columns
    constexpr int NAW = (TM + 3) / 4, NBW = TN;      // 1-KiB DMA pieces per wave and stage
    constexpr int VM_AT = (BM + BN) * 64;
    constexpr int STAGE = VM_AT + 1024;              // + 4 x 256 B of voxel masks (one copy per wave)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % prm.tiles_mn, slice = blockIdx.x / prm.tiles_mn;
    const int tm = tile % prm.tiles_m, tn = tile / prm.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int c0 = slice * prm.chunks_per_slice;
    const int nq = min(prm.chunks_per_slice, prm.chunks_total - c0);
    // Chunk order inside a clip.  Linear (walk_T = 0), or -- for kernels with temporal taps -- frame-minor:
    // the T frames of one 16-voxel (h,w) segment, then the next segment, so that the rows a tap reads at
    // frame t-1 / t+1 are the rows the neighbouring chunks just fetched (x then comes from HBM once, not kT times).
    const int cpc = prm.S / 16;                       // chunks per clip
    int n_img = c0 / cpc;
    int w_t = 0, w_seg = 0, p_local;
    if (prm.walk_T > 0) {
        const int cin = c0 - n_img * cpc;
        w_seg = cin / prm.walk_T;
        w_t = cin - w_seg * prm.walk_T;
        p_local = w_t * prm.HW + w_seg * 16;
    } else {
        p_local = (c0 - n_img * cpc) * 16;
    }

    // ---- DMA assignment: lane l of a piece fills row l/4, slot l%4 <- source slot (l%4) ^ swz(row) ----
    const int srcslot = ((lane & 3) ^ swz64(lane >> 2)) * 4;
    const float* zero = (const float*)zsv_wgrad_zero_line;
    const float* a_ptr[NAW];
    int a_step[NAW], a_dst[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        int pa = wave + 4 * k;
        if (pa >= TM) pa -= 4;                         // surplus slot: repeat this wave's previous piece
        const int m = m0 + pa * 16 + (lane >> 2);
        const bool ok = m < prm.M;
        a_ptr[k] = ok ? DY + ((size_t)n_img * prm.M + m) * prm.S + p_local + srcslot : zero;
        a_step[k] = ok ? 1 : 0;
        a_dst[k] = pa * 1024;
    }
    long b_off[NBW];
    int b_ok[NBW], b_dst[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int pb = wave + 4 * k;
        const int n = n0 + pb * 16 + (lane >> 2);
        const int tap = n / prm.Cpad, ci = n - tap * prm.Cpad;
        const int kt = tap / (prm.kH * prm.kW), r = tap - kt * prm.kH * prm.kW, kh = r / prm.kW, kw = r - kh * prm.kW;
        const int delta = (kt - prm.pT) * prm.HW + (kh - prm.pH) * prm.W + (kw - prm.pW);
        b_ok[k] = (n < prm.Kp && ci < prm.Cin) ? 1 : 0;
        b_off[k] = ((long)n_img * prm.Cin + ci) * prm.S + p_local + delta + srcslot;
        b_dst[k] = BM * 64 + pb * 1024;
    }
    const int vm_lane = lane & 15;
    // A 16-byte piece that straddles the first / last float of the tensor cannot be fetched whole: it is
    // DMA'd as zeros and its in-range floats are patched in afterwards (first and last chunks only).
    long patch_off[KB][NBW];
    bool patch[KB][NBW];

    auto issue = [&](int buf, int u) {
        unsigned char* base = lds + buf * STAGE;
#pragma unroll
        for (int k = 0; k < NAW; ++k) __builtin_amdgcn_global_load_lds(a_ptr[k], (lds_ptr_t)(base + a_dst[k]), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            const bool in = b_ok[k] && (unsigned long)b_off[k] < (unsigned long)(prm.x_elems - 3);
            patch[u][k] = b_ok[k] && !in && b_off[k] > -4 && b_off[k] < prm.x_elems;
            patch_off[u][k] = b_off[k];
            const float* src = in ? X + b_off[k] : zero;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(base + b_dst[k]), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds(VM + p_local + vm_lane, (lds_ptr_t)(base + VM_AT + wave * 256), 4, 0, 0);
        // next chunk: 16 voxels on (or the same segment one frame on), or the first voxels of the next clip
        int np;
        bool wrap = false;
        if (prm.walk_T > 0) {
            if (++w_t == prm.walk_T) {
                w_t = 0;
                if ((++w_seg) * 16 == prm.HW) { w_seg = 0; wrap = true; }
            }
            np = w_t * prm.HW + w_seg * 16;
        } else {
            np = p_local + 16;
            if (np == prm.S) { np = 0; wrap = true; }
        }
        long ja = np - p_local, jb = np - p_local;
        p_local = np;
        if (wrap) {
            ja += (long)prm.M * prm.S;
            jb += (long)prm.Cin * prm.S;
        }
#pragma unroll
        for (int k = 0; k < NAW; ++k) a_ptr[k] += a_step[k] ? ja : 0;
#pragma unroll
        for (int k = 0; k < NBW; ++k) b_off[k] += jb;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int g = lane >> 4, r16 = lane & 15;
    const unsigned frag = r16 * 64 + ((g ^ swz64(r16)) << 4);
    const unsigned a_frag = lds_base + frag;
    const unsigned b_frag = lds_base + BM * 64 + wave * TN * 1024 + frag;
    const unsigned v_frag = lds_base + VM_AT + wave * 256 + g * 16;
    int tap_of[TN];                                    // tap of each of this wave's 16-column blocks
#pragma unroll
    for (int j = 0; j < TN; ++j) tap_of[j] = min((n0 + (wave * TN + j) * 16) / prm.Cpad, 31);

    auto patch_edges = [&](int buf, int u) {            // after the chunk's DMAs have landed, before the barrier
        bool any = false;
#pragma unroll
        for (int k = 0; k < NBW; ++k) any = any || patch[u][k];
        if (__builtin_amdgcn_ballot_w64(any) == 0) return;
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            if (!patch[u][k]) continue;
            float* dst = (float*)(lds + buf * STAGE + b_dst[k] + lane * 16);
            for (int e = 0; e < 4; ++e)
                if ((unsigned long)(patch_off[u][k] + e) < (unsigned long)prm.x_elems) dst[e] = X[patch_off[u][k] + e];
        }
    };
    // groups of KB chunks: group gi lives in LDS buffers (gi & 1) * KB + u
    const int ng = (nq + KB - 1) / KB;
    auto issue_group = [&](int gi) {
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (gi * KB + u < nq) issue((gi & 1) * KB + u, u);
    };
    auto patch_group = [&](int gi) {
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (gi * KB + u < nq) patch_edges((gi & 1) * KB + u, u);
    };

    issue_group(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    patch_group(0);
    __builtin_amdgcn_s_barrier();
    for (int gi = 0; gi < ng; ++gi) {
        if (gi + 1 < ng) issue_group(gi + 1);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
        if (gi * KB + u >= nq) break;
        const unsigned so = ((gi & 1) * KB + u) * STAGE;
        // fragments: masks, B blocks, A0, A1 up front, then A(i+2) under the MFMAs of A(i)
        u32x4 vm;
        f32x4 bf[TN], af[3];
        lds_read128u(vm, v_frag + so);
#pragma unroll
        for (int j = 0; j < TN; ++j) lds_read128f<0>(bf[j], b_frag + so + j * 1024);
        lds_read128f<0>(af[0], a_frag + so);
        lds_read128f<1024>(af[1], a_frag + so);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i + 2 < TM) lds_read128f<0>(af[(i + 2) % 3], a_frag + so + (i + 2) * 1024);
            if (i == 0) {
                // everything but the two newest reads (A1, A2) has landed: masks and B blocks too
                lds_wait1<2>(af[0]);
                asm volatile("" : "+v"(vm));             // (the uses below must stay behind the wait)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bf[j]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bf[j][e] = ((vm[e] >> tap_of[j]) & 1u) ? bf[j][e] : 0.f;
            } else if (i + 2 < TM) lds_wait1<2>(af[i % 3]);
            else if (i + 1 < TM) lds_wait1<1>(af[i % 3]);
            else lds_wait1<0>(af[i % 3]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i % 3][s], bf[j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // group gi+1 has landed (issued a whole group ago)
        if (gi + 1 < ng) patch_group(gi + 1);
        __builtin_amdgcn_s_barrier();
    }

    // partial slab of this slice: OUT[slice][m][k'] (k' tap-major); lane holds rows 4g..4g+3 of column r16
    float* out = OUT + (size_t)slice * prm.M * prm.Kp;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int k = n0 + (wave * TN + j) * 16 + r16;
        if (k >= prm.Kp) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * i + 4 * g + r;
                if (m < prm.M) out[(size_t)m * prm.Kp + k] = acc[i][j][r];
            }
        }
    }
#endif
}

// ---- host side -----------------------------------------------------------------------------------
bool wgrad_dma_applicable(const zsv_conv_desc* d, const float* x, const float* dy) {
    if (ZSV_KNOB(NO_WGRAD_DMA)) return false;
    if (d->Cin < 16 || d->Cout < 16) return false;
    if (d->sT != 1 || d->sH != 1 || d->sW != 1) return false;
    if (d->To != d->Ti || d->Ho != d->Hi || d->Wo != d->Wi) return false;
    const long S = (long)d->Ti * d->Hi * d->Wi;
    if (S % 16 != 0 || d->kT * d->kH * d->kW > 32) return false;
    if (d->kT * d->kH * d->kW * ((d->Cin + 15) / 16 * 16) < 256) return false;     // few columns (T0: 144): measured slower
    if ((long)d->N * d->Cin * S >= (1L << 31) || (long)d->N * d->Cout * S >= (1L << 31)) return false;
    if (x != nullptr && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return false;
    return true;
}

struct WgradDmaPlan {
    int tm, tn, tiles_m, tiles_n, slices, chunks_per_slice, Cpad, Kp;
};

static WgradDmaPlan wgrad_dma_plan(const zsv_conv_desc* d) {
    WgradDmaPlan pl;
    const int M = d->Cout, taps = d->kT * d->kH * d->kW;
    pl.Cpad = (d->Cin + 15) / 16 * 16;
    pl.Kp = taps * pl.Cpad;
    const long chunks = (long)d->N * d->Ti * d->Hi * d->Wi / 16;
    // (rows, columns) tile = least padded MACs x shape factor / round fill, as in conv_wgrad.hip; the
    // column tiles 64 / 128 / 192 exist because Kp = taps * Cpad is a multiple of 64 or 192 far more
    // often than of 128 (576 = 3 x 192, 432 -> 448 = 7 x 64)
    const int tms[4] = {9, 8, 4, 5};
    const double pen_m[4] = {1.00, 1.00, 1.05, 1.08};
    const int tns[4] = {2, 3, 1, 7};                    // 7: a 64-row problem with <= 448 columns in ONE tile (dY read once)
    const double pen_n[4] = {1.00, 1.00, 1.06, 1.00};
    long max_slices = (chunks * 16 + 511) / 512;         // at least 512 voxels per slice
    if (max_slices < 1) max_slices = 1;
    if (max_slices > 1024) max_slices = 1024;
    double best_w = 1e300;
    pl.tm = 9; pl.tn = 2; pl.slices = 1;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const int bm = 16 * tms[i], bn = 64 * tns[j];
            if (tms[i] * tns[j] > 28) continue;          // accumulator registers
            const long tm_ = (M + bm - 1) / bm, tn_ = (pl.Kp + bn - 1) / bn, tiles = tm_ * tn_;
            const long lds = 2L * (tms[i] * tns[j] <= 10 ? 2 : 1) * ((bm + bn) * 64 + 1024);
            long per_cu = (160L * 1024) / lds;
            const long by_regs = tms[i] * tns[j] > 27 ? 2 : 3;     // __launch_bounds__
            if (per_cu > by_regs) per_cu = by_regs;
            const long resident = 256L * per_cu;
            long sl = 1;
            double best_eff = -1.0;
            for (long s = 1; s <= max_slices && tiles * s <= 4 * resident + tiles; ++s) {
                const long wgs = tiles * s, rounds = (wgs + resident - 1) / resident;
                const double eff = (double)wgs / (double)(rounds * resident) - 0.02 * (double)wgs / (double)resident;
                if (eff > best_eff + 1e-9) { best_eff = eff; sl = s; }
            }
            const double w = (double)(tm_ * bm) * (double)(tn_ * bn) * pen_m[i] * pen_n[j] / (best_eff > 1e-3 ? best_eff : 1e-3);
            if (w < best_w * 0.999) { best_w = w; pl.tm = tms[i]; pl.tn = tns[j]; pl.slices = (int)sl; }
        }
    if (const char* e = ZSV_KNOB(WGRAD_DMA_TM)) { const int t = atoi(e); if (t == 9 || t == 8 || t == 4 || t == 5) pl.tm = t; }
    if (const char* e = ZSV_KNOB(WGRAD_DMA_TN)) { const int t = atoi(e); if ((t >= 1 && t <= 3 && t * pl.tm <= 27) || (t == 7 && pl.tm == 4)) pl.tn = t; }
    const int bm = 16 * pl.tm, bn = 64 * pl.tn;
    pl.tiles_m = (M + bm - 1) / bm;
    pl.tiles_n = (pl.Kp + bn - 1) / bn;
    if (const char* e = ZSV_KNOB(WGRAD_DMA_SLICES)) pl.slices = atoi(e);
    if (pl.slices > max_slices) pl.slices = (int)max_slices;
    if (pl.slices < 1) pl.slices = 1;
    pl.chunks_per_slice = (int)((chunks + pl.slices - 1) / pl.slices);
    pl.slices = (int)((chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice);
    return pl;
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t wgrad_dma_workspace_bytes(const zsv_conv_desc* d) {
    const WgradDmaPlan pl = wgrad_dma_plan(d);
    const size_t S = (size_t)d->Ti * d->Hi * d->Wi;
    return align256((size_t)pl.slices * d->Cout * pl.Kp * sizeof(float)) + S * sizeof(unsigned);
}

template <int TM, int TN>
static int wgrad_dma_launch(const WgradDmaParams& p, int slices, hipStream_t stream, const float* x, const float* dy,
                            const unsigned* vm, float* out) {
    constexpr int KB = TM * TN <= 10 ? 2 : 1;
    constexpr int LDS_BYTES = 2 * KB * ((16 * TM + 64 * TN) * 64 + 1024);
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<TM, TN, KB>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<TM, TN, KB>), dim3((unsigned)(p.tiles_mn * slices)), dim3(256), LDS_BYTES,
                       stream, p, x, dy, vm, out);
    return launch_status();
}

template <int TM>
static int wgrad_dma_launch_tn(int tn, const WgradDmaParams& p, int slices, hipStream_t stream, const float* x,
                               const float* dy, const unsigned* vm, float* out) {
    if (tn == 1) return wgrad_dma_launch<TM, 1>(p, slices, stream, x, dy, vm, out);
    if constexpr (TM == 4) {
        if (tn == 7) return wgrad_dma_launch<TM, 7>(p, slices, stream, x, dy, vm, out);
    }
    if constexpr (TM * 3 <= 27) {
        if (tn == 3) return wgrad_dma_launch<TM, 3>(p, slices, stream, x, dy, vm, out);
    }
    return wgrad_dma_launch<TM, 2>(p, slices, stream, x, dy, vm, out);
}

int wgrad_vmask(const zsv_conv_desc* d, unsigned* out, hipStream_t stream) {
    const int S = d->Ti * d->Hi * d->Wi;
    hipLaunchKernelGGL(wgrad_vmask_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream, S, d->Ti, d->Hi, d->Wi,
                       d->kT, d->kH, d->kW, d->pT, d->pH, d->pW, out);
    return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
}

// slabs + mask table live in `workspace`; returns the slab geometry for slab_sum_kernel
int wgrad_dma(const zsv_conv_desc* d, const float* x, const float* dy, void* workspace, size_t workspace_bytes,
              int* slices_out, int* cpad_out, hipStream_t stream, const unsigned* vm_ext) {
    const WgradDmaPlan pl = wgrad_dma_plan(d);
    if (!workspace || workspace_bytes < wgrad_dma_workspace_bytes(d)) return ZSV_E_WORKSPACE;
    WgradDmaParams p;
    p.M = d->Cout; p.Cin = d->Cin; p.Cpad = pl.Cpad; p.taps = d->kT * d->kH * d->kW; p.Kp = pl.Kp;
    p.S = d->Ti * d->Hi * d->Wi; p.HW = d->Hi * d->Wi; p.W = d->Wi;
    p.kH = d->kH; p.kW = d->kW; p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
    p.chunks_total = (int)((long)d->N * p.S / 16);
    p.walk_T = (d->kT > 1 && d->Ti > 1 && p.HW % 16 == 0 && ZSV_KNOB(WGRAD_LINEAR_WALK) == nullptr) ? d->Ti : 0;
    p.chunks_per_slice = pl.chunks_per_slice;
    p.x_elems = (long)d->N * d->Cin * p.S;
    p.tiles_m = pl.tiles_m; p.tiles_mn = pl.tiles_m * pl.tiles_n;
    float* slabs = (float*)workspace;
    const unsigned* vm = vm_ext;                       // (kept by the caller: zsv_conv3d_wgrad_masked)
    int st = ZSV_OK;
    if (vm == nullptr) {
        unsigned* own = (unsigned*)((char*)workspace + align256((size_t)pl.slices * d->Cout * pl.Kp * sizeof(float)));
        st = wgrad_vmask(d, own, stream);
        if (st) return st;
        vm = own;
    }
    switch (pl.tm) {
        case 9: st = wgrad_dma_launch_tn<9>(pl.tn, p, pl.slices, stream, x, dy, vm, slabs); break;
        case 8: st = wgrad_dma_launch_tn<8>(pl.tn, p, pl.slices, stream, x, dy, vm, slabs); break;
        case 5: st = wgrad_dma_launch_tn<5>(pl.tn, p, pl.slices, stream, x, dy, vm, slabs); break;
        default: st = wgrad_dma_launch_tn<4>(pl.tn, p, pl.slices, stream, x, dy, vm, slabs); break;
    }
    *slices_out = pl.slices;
    *cpad_out = pl.Cpad;
    return st;
}

}  // namespace zsv
