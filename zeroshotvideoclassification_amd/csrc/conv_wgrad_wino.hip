// conv_wgrad_wino.hip -- weight gradient of 1x3x3 / 3x3x3 stride-1 "same" convolutions with the kw taps in
// Winograd form (the transpose of F(2,3) along W), fp32.
//
//   dW[co][ci][r][kw] = sum_p dY[co][p] * X[ci][p + d(r) + kw - 1]        r = (kt, kh) row tap, d(r) its voxel shift
// (aten::convolution_backward, weight part, for resnet.py:23-30,40-45 and network.py:102-117).  Take the output voxels in
// pairs (2q, 2q+1) along W -- W is even, a pair never straddles a row -- with y0, y1 = dY at the pair and d0..d3 =
// X[2q-1 .. 2q+2] of one (ci, r) row (zero outside the image):
//     A = (y0, y0+y1, y0-y1, y1)        V = (d0-d2, d1+d2, d2-d1, d1-d3)        Mi = sum_q Ai * Vi
//     dW[kw=0] = M0 + (M1+M2)/2     dW[kw=1] = (M1-M2)/2     dW[kw=2] = (M1+M2)/2 - M3
// 4 multiplies per pair and (co, ci, r) instead of 6: 1.5x fewer MFMAs than conv_wgrad_dma.hip, still fp32 in / fp32
// accumulate (V is the same transform as the forward's, conv_wino.hip).
//
// GEMM per point: rows = 16*TM output channels, columns = 64*TN (r, ci) pairs per workgroup (TN 16-column blocks per wave:
// 1 for the 128 / 144-row tiles, 2 for the 64-row tile of the 64-channel layers),
// reduction over voxel pairs, cut into slices (one round of workgroups covers the problem); every slice writes its
// partial [point][co][(r, ci)] slab and wgrad_wino_sum_kernel adds the slabs in a fixed order and applies the output
// transform: bitwise reproducible, no float atomics.
//   * MFMA k order as in conv_wgrad_dma.hip: k = 4*(lane>>4) + step, so a lane needs 4 consecutive PAIRS = 8 voxels
//     of its row for the 4 steps of a 32-voxel chunk: two ds_read_b128 per fragment (+ the two halo voxels of the X row).
//   * Both operands are rows contiguous along the reduction and are staged by 16-byte LDS-DMAs (buffer_load_dwordx4 ... lds:
//     a piece outside the tensor reads as zeros) into padded rows: dY 36 floats (32 + one dummy piece), X 44 floats
//     (voxels p-4 .. p+39 of the shifted row) -- pitches of 36 / 44 banks make the ds_read_b128 fragments conflict-free.
//   * Borders: the per-voxel tap-validity words of conv_wgrad_dma.hip (wgrad_vmask_kernel) are DMA'd with the operands;
//     d0 / (d1, d2) / d3 of a pair are zeroed by the bits (r, kw=0) / (r, kw=1) of its first and (r, kw=2) of its second
//     voxel.  Rows read past a row / plane / clip pick up neighbouring values, which the same bits discard.  A clip whose
//     voxel count is not a multiple of 32 ends with a partial chunk: its dY pieces past the clip are DMA'd as zeros and the
//     mask words past the clip read as zero.  X pieces need not be 16-byte aligned (W % 4 != 0), only dword aligned.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "conv_params.h"
#include "zsv_common.h"
#include "zsv_hip.h"
#include "knobs.h"

namespace zsv {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

struct WgradWinoParams {
    int M, Cin, Cpad, R, Kp;          // Kp = R * Cpad columns, row-tap major
    int S, HW, W, kT;
    int chunks_total, chunks_per_slice, cpc; // 32-voxel chunks; cpc = chunks per clip (the last one may be partly past the clip)
    unsigned x_bytes, dy_bytes, vm_bytes;
    int tiles_m, tiles_mn;
};

// EDGE = the clip's voxel count is not a multiple of 32 or W is not a multiple of 4 (partial last chunk, unaligned X pieces)
template <int TM, int TN, bool EDGE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_wino_kernel(WgradWinoParams prm, const float* __restrict__ X,
                                                                 const float* __restrict__ DY,
                                                                 const unsigned* __restrict__ VM, float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 16 * TM;
    constexpr int LDA = 36, LDB = 44;                       // floats per LDS row (9 / 11 pieces of 16 bytes)
    constexpr int NA_TOT = (BM * 9 + 63) / 64;              // 1-KiB DMA instructions per stage: dY rows
    constexpr int BN = 64 * TN;                             // columns per workgroup
    constexpr int NB_TOT = 11 * TN;                         //                                    X rows (BN x 11 pieces)
    constexpr int NA = (NA_TOT + 3) / 4, NB = (NB_TOT + 3) / 4;   // per wave
    constexpr int A_BYTES = NA_TOT * 1024, B_BYTES = NB_TOT * 1024;
    constexpr int VM_AT = A_BYTES + B_BYTES;
    constexpr int STAGE = VM_AT + 1024;
    constexpr unsigned OOB = 0xFFFFFFF0u;                   // (+12 must not wrap)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the column / row tiles of one slice read the same dY (X) rows: consecutive logical ids share an XCD (one L2)
    const int lid = xcd_tile(gridDim.x, blockIdx.x);
    const int tile = lid % prm.tiles_mn, slice = lid / prm.tiles_mn;
    const int m0 = (tile % prm.tiles_m) * BM, n0 = (tile / prm.tiles_m) * BN;
    const int c0 = slice * prm.chunks_per_slice;
    const int nq = min(prm.chunks_per_slice, prm.chunks_total - c0);
    const int cpc = prm.cpc;
    int n_img = c0 / cpc;
    int p_local = (c0 - n_img * cpc) * 32;

    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, prm.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_vm = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(VM), 0, prm.vm_bytes, 0x00020000);

    // ---- DMA assignment: piece id = 64 * instruction + lane -> (row, 16-byte piece of the row) ------------------
    int a_off[NA], a_vox[NA];           // byte offset inside clip 0 at p_local = 0 (or -1: reads as zeros); first voxel of the piece
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int id = 64 * (wave + 4 * k) + lane;
        const int row = id / 9, pc = id - row * 9;
        const bool ok = row < BM && m0 + row < prm.M && pc < 8;
        a_off[k] = ok ? 4 * ((m0 + row) * prm.S + 4 * pc) : -1;
        a_vox[k] = 4 * pc;
    }
    int b_off[NB];
    bool b_ok[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int id = 64 * (wave + 4 * k) + lane;
        const int row = id / 11, pc = id - row * 11;
        const int n = n0 + row;
        const int r = n / prm.Cpad, ci = n - r * prm.Cpad;
        const int kt = r / 3, kh = r - 3 * kt;
        const int delta = (kt - prm.kT / 2) * prm.HW + (kh - 1) * prm.W;
        b_ok[k] = row < BN && n < prm.Kp && ci < prm.Cin;
        b_off[k] = 4 * (ci * prm.S + delta + 4 * pc - 4);          // may be negative at p_local = 0 (then out of range)
    }

    // A 16-byte X piece that straddles the first float of the tensor (only with W % 4 != 0: clip 0, channel 0, first chunk)
    // cannot be fetched whole: it is DMA'd as zeros and its in-range floats are patched in after the wait.
    int patch_off[NB];
    auto issue = [&](int buf) {
        unsigned char* base = lds + buf * STAGE;
        const int a_chunk = 4 * (n_img * prm.M * prm.S + p_local);
        const int b_chunk = 4 * (n_img * prm.Cin * prm.S + p_local);
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (wave + 4 * k < NA_TOT)                              // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(base + 1024 * (wave + 4 * k)), 16,
                                                         (int)((a_off[k] >= 0 && (!EDGE || p_local + a_vox[k] < prm.S)) ? (unsigned)(a_off[k] + a_chunk) : OOB), 0, 0, 0);   // (dY past the clip = 0)
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (wave + 4 * k < NB_TOT) {
                const int off = b_off[k] + b_chunk;                 // negative = before the tensor: reads as zeros
                if constexpr (EDGE) patch_off[k] = (b_ok[k] && off < 0 && off > -16) ? off : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(base + A_BYTES + 1024 * (wave + 4 * k)), 16,
                                                         (int)((b_ok[k] && off >= 0) ? (unsigned)off : OOB), 0, 0, 0);
            }
        if (wave == 0)                                              // 32 mask words: lanes 0..7
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_vm, (lds_ptr_t)(base + VM_AT), 16,
                                                     (int)(lane < 8 ? (unsigned)(4 * (p_local + 4 * lane)) : OOB), 0, 0, 0);
        p_local += 32;
        if (p_local >= prm.S) { p_local = 0; ++n_img; }
    };

    auto patch_edges = [&](int buf) {                   // after the chunk's DMAs have landed, before the barrier
        if constexpr (!EDGE) return;
        bool any = false;
#pragma unroll
        for (int k = 0; k < NB; ++k) any = any || (wave + 4 * k < NB_TOT && patch_off[k] != 0);
        if (__builtin_amdgcn_ballot_w64(any) == 0) return;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (wave + 4 * k >= NB_TOT || patch_off[k] == 0) continue;
            float* dst = (float*)(lds + buf * STAGE + A_BYTES + 1024 * (wave + 4 * k) + lane * 16);
            for (int e = 0; e < 4; ++e)
                if (patch_off[k] + 4 * e >= 0) dst[e] = X[(patch_off[k] + 4 * e) / 4];
        }
    };

    f32x4 acc[4][TM][TN];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    unsigned bit0[TN], bit1[TN], bit2[TN];                            // row tap of each of this wave's 16-column blocks (Cpad % 16 == 0)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r_tap = min((n0 + 16 * (wave * TN + j)) / prm.Cpad, 8);
        bit0[j] = 1u << (3 * r_tap); bit1[j] = 2u << (3 * r_tap); bit2[j] = 4u << (3 * r_tap);
    }
    const int a_frag = (r16 * LDA + 8 * g) * 4;                       // bytes inside the A image (+ 16 * LDA * 4 per block)
    const int b_frag = A_BYTES + ((16 * wave * TN + r16) * LDB + 8 * g + 4) * 4;      // (+ 16 * LDB * 4 per column block)
    const int v_frag = VM_AT + g * 32;

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    patch_edges(0);
    __syncthreads();
    for (int ch = 0; ch < nq; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nq) issue(cur ^ 1);
        const unsigned char* st = lds + cur * STAGE;
        // this lane's 8 voxels (4 pairs) of its X row, their halo and their mask words
        const u32x4w vm0 = *reinterpret_cast<const u32x4w*>(st + v_frag), vm1 = *reinterpret_cast<const u32x4w*>(st + v_frag + 16);
        f32x4 ar[2][2];
        ar[0][0] = *reinterpret_cast<const f32x4*>(st + a_frag);
        ar[0][1] = *reinterpret_cast<const f32x4*>(st + a_frag + 16);
        const unsigned mw[8] = {vm0[0], vm0[1], vm0[2], vm0[3], vm1[0], vm1[1], vm1[2], vm1[3]};
        float V[TN][4][4];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned char* bp = st + b_frag + j * 16 * LDB * 4;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 16);
            const float bl = *reinterpret_cast<const float*>(bp - 4), br = *reinterpret_cast<const float*>(bp + 32);
            const float xv[10] = {bl, b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], br};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned ma = mw[2 * s], mb = mw[2 * s + 1];
                const float d0 = (ma & bit0[j]) ? xv[2 * s] : 0.f, d1 = (ma & bit1[j]) ? xv[2 * s + 1] : 0.f,
                            d2 = (ma & bit1[j]) ? xv[2 * s + 2] : 0.f, d3 = (mb & bit2[j]) ? xv[2 * s + 3] : 0.f;
                V[j][0][s] = d0 - d2; V[j][1][s] = d1 + d2; V[j][2][s] = d2 - d1; V[j][3][s] = d1 - d3;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int sl = i & 1;
            if (i + 1 < TM) {
                ar[sl ^ 1][0] = *reinterpret_cast<const f32x4*>(st + a_frag + (i + 1) * 16 * LDA * 4);
                ar[sl ^ 1][1] = *reinterpret_cast<const f32x4*>(st + a_frag + (i + 1) * 16 * LDA * 4 + 16);
            }
            const float y[8] = {ar[sl][0][0], ar[sl][0][1], ar[sl][0][2], ar[sl][0][3], ar[sl][1][0], ar[sl][1][1], ar[sl][1][2], ar[sl][1][3]};
            float sum[4], dif[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { sum[s] = y[2 * s] + y[2 * s + 1]; dif[s] = y[2 * s] - y[2 * s + 1]; }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[2 * s], V[j][0][s], acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sum[s], V[j][1][s], acc[1][i][j], 0, 0, 0);
                    acc[2][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dif[s], V[j][2][s], acc[2][i][j], 0, 0, 0);
                    acc[3][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[2 * s + 1], V[j][3][s], acc[3][i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ch + 1 < nq) patch_edges(cur ^ 1);
        __syncthreads();
    }

    // partial slab of this slice: OUT[slice][point][m][n]; lane holds rows 4g..4g+3 of column r16
    float* out = OUT + (size_t)slice * 4 * prm.M * prm.Kp;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + 16 * (wave * TN + j) + r16;
        if (n >= prm.Kp) continue;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * i + 4 * g + r;
                    if (m < prm.M) out[((size_t)p * prm.M + m) * prm.Kp + n] = acc[p][i][j][r];
                }
    }
#endif
}

// ---- the same weight gradient with the transpose of F(4,3) along W (round 4) ---------------------------------------
// Output voxels in QUADS (4q .. 4q+3; W % 4 == 0, a quad never straddles a row), y0..y3 = dY at the quad, d0..d5 =
// X[4q-1 .. 4q+4] of one (ci, r) row (zero outside the image).  With A, B^T, G of F(4,3) (Lavin & Gray; the forward of
// conv_wino.hip uses the same B^T):
//     A = (y0, e+o, e-o, e4+2q, e4-2q, y3)     e = y0+y2, o = y1+y3, e4 = y0+4y2, q = y1+4y3
//     V = B^T d                                 Mi = sum_quads Ai * Vi                dW = G^T M
// 6 multiplies per quad and (co, ci, r) instead of 8 for its two pairs: a quarter of the MFMAs of the F(2,3) form gone.
// Six accumulator sets do not fit the 144-row tile of a four-wave workgroup at two workgroups per CU (216 + ~90 registers
// against 256), and an 80-row tile would stage 1.8x the X bytes per MFMA.  So: ONE workgroup of EIGHT waves per CU, same
// LDS image as before (144 dY rows, 64 X columns per chunk), waves 0-3 take the first TMA row blocks, waves 4-7 the other
// TMB (5 + 4 = the nine blocks of 144 rows, 4 + 4 for the 128-row tile): two waves per SIMD as before, 120 accumulator
// registers per wave, and the X image is staged once per 144 rows.
template <int TMA, int TMB>
__global__ __launch_bounds__(512) void conv_wgrad_wino4_kernel(WgradWinoParams prm, const float* __restrict__ X,
                                                               const float* __restrict__ DY, const unsigned* __restrict__ VM,
                                                               float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TMX = TMA > TMB ? TMA : TMB;
    constexpr int BM = 16 * (TMA + TMB);
    constexpr int LDA = 36, LDB = 44;
    constexpr int NA_TOT = (BM * 9 + 63) / 64, NB_TOT = 11;
    constexpr int NA = (NA_TOT + 7) / 8, NB = (NB_TOT + 7) / 8;     // per wave (8 waves)
    constexpr int A_BYTES = NA_TOT * 1024, B_BYTES = NB_TOT * 1024;
    constexpr int VM_AT = A_BYTES + B_BYTES;
    constexpr int STAGE = VM_AT + 1024;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, cw = wave & 3;                      // row half, 16-column block
    const int i0 = half ? TMA : 0, tmw = half ? TMB : TMA;
    const int lid = xcd_tile(gridDim.x, blockIdx.x);
    const int tile = lid % prm.tiles_mn, slice = lid / prm.tiles_mn;
    const int m0 = (tile % prm.tiles_m) * BM, n0 = (tile / prm.tiles_m) * 64;
    const int c0 = slice * prm.chunks_per_slice;
    const int nq = min(prm.chunks_per_slice, prm.chunks_total - c0);
    int n_img = c0 / prm.cpc;
    int p_local = (c0 - n_img * prm.cpc) * 32;

    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, prm.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_vm = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(VM), 0, prm.vm_bytes, 0x00020000);

    int a_off[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int id = 64 * (wave + 8 * k) + lane;
        const int row = id / 9, pc = id - row * 9;
        const bool ok = row < BM && m0 + row < prm.M && pc < 8;
        a_off[k] = ok ? 4 * ((m0 + row) * prm.S + 4 * pc) : -1;
    }
    int b_off[NB];
    bool b_ok[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int id = 64 * (wave + 8 * k) + lane;
        const int row = id / 11, pc = id - row * 11;
        const int n = n0 + row;
        const int r = n / prm.Cpad, ci = n - r * prm.Cpad;
        const int kt = r / 3, kh = r - 3 * kt;
        const int delta = (kt - prm.kT / 2) * prm.HW + (kh - 1) * prm.W;
        b_ok[k] = row < 64 && n < prm.Kp && ci < prm.Cin;
        b_off[k] = 4 * (ci * prm.S + delta + 4 * pc - 4);
    }

    auto issue = [&](int buf) {
        unsigned char* base = lds + buf * STAGE;
        const int a_chunk = 4 * (n_img * prm.M * prm.S + p_local);
        const int b_chunk = 4 * (n_img * prm.Cin * prm.S + p_local);
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (wave + 8 * k < NA_TOT)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(base + 1024 * (wave + 8 * k)), 16,
                                                         (int)(a_off[k] >= 0 ? (unsigned)(a_off[k] + a_chunk) : OOB), 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (wave + 8 * k < NB_TOT) {
                const int off = b_off[k] + b_chunk;                 // negative = before the tensor: reads as zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(base + A_BYTES + 1024 * (wave + 8 * k)), 16,
                                                         (int)((b_ok[k] && off >= 0) ? (unsigned)off : OOB), 0, 0, 0);
            }
        if (wave == 7)                                              // 32 mask words: lanes 0..7 (wave 7 has the fewest pieces)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_vm, (lds_ptr_t)(base + VM_AT), 16,
                                                     (int)(lane < 8 ? (unsigned)(4 * (p_local + 4 * lane)) : OOB), 0, 0, 0);
        p_local += 32;
        if (p_local >= prm.S) { p_local = 0; ++n_img; }
    };

    f32x4 acc[6][TMX];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int i = 0; i < TMX; ++i) acc[p][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    const int r_tap = min((n0 + 16 * cw) / prm.Cpad, 8);            // row tap of this wave's 16 columns (Cpad % 16 == 0)
    const unsigned bit0 = 1u << (3 * r_tap), bit1 = 2u << (3 * r_tap), bit2 = 4u << (3 * r_tap);
    const int a_frag = ((16 * i0 + r16) * LDA + 8 * g) * 4;
    const int b_frag = A_BYTES + ((16 * cw + r16) * LDB + 8 * g + 4) * 4;
    const int v_frag = VM_AT + g * 32;

    // (Tried on top of this, each A/B-ed on one device: requests two and three chunks ahead (3 / 4 LDS stages) -- slower, the
    // chunk-end wait is not what the loop loses; the barrier in front of the chunk's last row block with the next chunk's
    // loads and V transform behind it -- S1 -3 % instead of -7 %.  Removing work shows where the time is
    // (profiles/r04_wgrad_f43_ablation.txt): DMA issue 9 %, barrier 6 %, V transform + its loads 11 %, the rest is the
    // dY transform and the MFMAs' own issue: the F(4,3) form trades a quarter of the MFMAs for 1.75x the VALU work per MFMA.)
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = 0; ch < nq; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nq) issue(cur ^ 1);
        const unsigned char* st = lds + cur * STAGE;
        const u32x4w vm0 = *reinterpret_cast<const u32x4w*>(st + v_frag), vm1 = *reinterpret_cast<const u32x4w*>(st + v_frag + 16);
        f32x4 ar[2][2];
        ar[0][0] = *reinterpret_cast<const f32x4*>(st + a_frag);
        ar[0][1] = *reinterpret_cast<const f32x4*>(st + a_frag + 16);
        float V[6][2];
        {
            const unsigned char* bp = st + b_frag;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 16);
            const float bl = *reinterpret_cast<const float*>(bp - 4), br = *reinterpret_cast<const float*>(bp + 32);
            const float xv[10] = {bl, b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], br};
            const unsigned mfirst[2] = {vm0[0], vm1[0]}, mlast[2] = {vm0[3], vm1[3]};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool row_in = (mfirst[s] & bit1) != 0;        // the four voxels of a quad share (t, h): one bit for d1..d4
                const float d0 = (mfirst[s] & bit0) ? xv[4 * s] : 0.f, d1 = row_in ? xv[4 * s + 1] : 0.f,
                            d2 = row_in ? xv[4 * s + 2] : 0.f, d3 = row_in ? xv[4 * s + 3] : 0.f,
                            d4 = row_in ? xv[4 * s + 4] : 0.f, d5 = (mlast[s] & bit2) ? xv[4 * s + 5] : 0.f;
                const float t5 = d3 - d1, t6 = d4 - d2;
                V[0][s] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                V[1][s] = fmaf(-4.f, d1 + d2, d3 + d4);
                V[2][s] = fmaf(4.f, d1 - d2, d4 - d3);
                V[3][s] = fmaf(2.f, t5, t6);
                V[4][s] = fmaf(-2.f, t5, t6);
                V[5][s] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
            }
        }
#pragma unroll
        for (int i = 0; i < TMX; ++i) {
            if (TMA != TMB && i >= tmw) break;                      // wave-uniform
            const int sl = i & 1;
            if (i + 1 < TMX) {                                      // (one block past a short half reads rows of the other half: unused)
                ar[sl ^ 1][0] = *reinterpret_cast<const f32x4*>(st + a_frag + (i + 1) * 16 * LDA * 4);
                ar[sl ^ 1][1] = *reinterpret_cast<const f32x4*>(st + a_frag + (i + 1) * 16 * LDA * 4 + 16);
            }
            float Av[6][2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float y0 = ar[sl][s][0], y1 = ar[sl][s][1], y2 = ar[sl][s][2], y3 = ar[sl][s][3];
                const float e = y0 + y2, o = y1 + y3, e4 = fmaf(4.f, y2, y0), q = fmaf(4.f, y3, y1);
                Av[0][s] = y0; Av[1][s] = e + o; Av[2][s] = e - o;
                Av[3][s] = fmaf(2.f, q, e4); Av[4][s] = fmaf(-2.f, q, e4); Av[5][s] = y3;
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 6; ++p)
                    acc[p][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(Av[p][s], V[p][s], acc[p][i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // partial slab of this slice: OUT[slice][point][m][n]; lane holds rows 4g..4g+3 of column r16
    float* out = OUT + (size_t)slice * 6 * prm.M * prm.Kp;
    const int n = n0 + 16 * cw + r16;
    if (n < prm.Kp) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int i = 0; i < TMX; ++i) {
                if (TMA != TMB && i >= tmw) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * (i0 + i) + 4 * g + r;
                    if (m < prm.M) out[((size_t)p * prm.M + m) * prm.Kp + n] = acc[p][i][r];
                }
            }
    }
#endif
}

// dW[co][ci][r][kw] from the slabs [slice][point][co][r * Cpad + ci]: 32 elements x 8 slice groups per block (group g adds
// slices g, g+8, ... in order, then the groups are added in order -- fixed order, bitwise reproducible), then G^T.
template <int PTS>
__global__ __launch_bounds__(256) void wgrad_wino_sum_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int M,
                                                             int Cin, int R, int Cpad, int slices) {
    __shared__ float part[8][PTS][32];
    const size_t plane = (size_t)M * R * Cpad;              // one point of one slice
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (size_t j0 = (size_t)blockIdx.x * 32; j0 < plane; j0 += (size_t)gridDim.x * 32) {
        const size_t j = j0 + e;
        const int ci = (int)(j % Cpad);
        const bool live = j < plane && ci < Cin;
        float s[PTS];
#pragma unroll
        for (int p = 0; p < PTS; ++p) s[p] = 0.f;
        if (live) {
            int k = grp;
            for (; k + 8 < slices; k += 16) {                     // two slices' loads in flight, added in the same order
                float v[2][PTS];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int p = 0; p < PTS; ++p) v[u][p] = slabs[((size_t)(k + 8 * u) * PTS + p) * plane + j];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int p = 0; p < PTS; ++p) s[p] += v[u][p];
            }
            for (; k < slices; k += 8)
#pragma unroll
                for (int p = 0; p < PTS; ++p) s[p] += slabs[((size_t)k * PTS + p) * plane + j];
        }
#pragma unroll
        for (int p = 0; p < PTS; ++p) part[grp][p][e] = s[p];
        __syncthreads();
        if (grp == 0 && live) {
            float Mv[PTS];
#pragma unroll
            for (int p = 0; p < PTS; ++p) {
                float t = part[0][p][e];
#pragma unroll
                for (int q = 1; q < 8; ++q) t += part[q][p][e];
                Mv[p] = t;
            }
            const size_t rr = j / Cpad;
            const int r = (int)(rr % R), co = (int)(rr / R);
            float* o = dw + (((size_t)co * Cin + ci) * R + r) * 3;
            if constexpr (PTS == 4) {
                const float h = 0.5f * (Mv[1] + Mv[2]);
                o[0] = Mv[0] + h;
                o[1] = 0.5f * (Mv[1] - Mv[2]);
                o[2] = h - Mv[3];
            } else {                                                // G^T of F(4,3)
                const float s12 = Mv[1] + Mv[2], d12 = Mv[2] - Mv[1], s34 = Mv[3] + Mv[4], d34 = Mv[3] - Mv[4];
                o[0] = 0.25f * Mv[0] - s12 * (1.f / 6.f) + s34 * (1.f / 24.f);
                o[1] = d12 * (1.f / 6.f) + d34 * (1.f / 12.f);
                o[2] = Mv[5] - s12 * (1.f / 6.f) + s34 * (1.f / 6.f);
            }
        }
        __syncthreads();
    }
}

// ---- host side -----------------------------------------------------------------------------------
struct WgradWinoPlan {
    int tm, tn, tiles_m, tiles_n, slices, chunks_per_slice, Cpad, Kp;
};

static WgradWinoPlan wgrad_wino_plan(const zsv_conv_desc* d) {
    WgradWinoPlan pl;
    const int M = d->Cout;
    pl.Cpad = (d->Cin + 15) / 16 * 16;
    pl.Kp = 3 * d->kT * pl.Cpad;
    const int p9 = (M + 143) / 144 * 144, p8 = (M + 127) / 128 * 128;
    pl.tm = M <= 64 ? 4 : (p9 <= p8 ? 9 : 8);
    pl.tn = pl.tm == 4 ? 2 : 1;
    const int bm = 16 * pl.tm, bn = 64 * pl.tn;
    pl.tiles_m = (M + bm - 1) / bm;
    pl.tiles_n = (pl.Kp + bn - 1) / bn;
    const long chunks = (long)d->N * (((long)d->Ti * d->Hi * d->Wi + 31) / 32);
    const long tiles = (long)pl.tiles_m * pl.tiles_n, resident = 512;       // 2 workgroups per CU
    // slices: the count that minimises (MFMA time / fill of the rounds of resident workgroups) + (slab write + read),
    // at least 24 chunks (768 voxels) per slice
    const double voxels = (double)chunks * 32.0;
    const double t_mfma = 4.0 * (double)(pl.tiles_m * bm) * (double)(pl.tiles_n * bn) * voxels / 1.1e14;
    const double t_slice = 2.0 * 4.0 * (double)M * pl.Kp * sizeof(float) / 6.0e12;
    long max_sl = chunks / 24;
    if (max_sl < 1) max_sl = 1;
    if (max_sl > 4096) max_sl = 4096;
    long sl = 1;
    double best = 1e300;
    for (long c = 1; c <= max_sl; ++c) {
        const long wgs = tiles * c, rounds = (wgs + resident - 1) / resident;
        const double cost = t_mfma * (double)(rounds * resident) / (double)wgs + t_slice * (double)c;
        if (cost < best * 0.999) { best = cost; sl = c; }
    }
    if (const char* e = ZSV_KNOB(WGRAD_WINO_SLICES)) sl = atol(e);
    if (sl < 1) sl = 1;
    if (sl > chunks) sl = chunks;
    pl.chunks_per_slice = (int)((chunks + sl - 1) / sl);
    pl.slices = (int)((chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice);
    return pl;
}

bool wgrad_wino_applicable(const zsv_conv_desc* d, const float* x, const float* dy) {
    if (ZSV_KNOB(NO_WINO) || ZSV_KNOB(NO_WGRAD_WINO)) return false;
    if ((d->kT != 1 && d->kT != 3) || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT != d->kT / 2 ||
        d->pH != 1 || d->pW != 1)
        return false;
    if (d->Wi % 2 != 0 || d->Cin < 16) return false;
    const long S = (long)d->Ti * d->Hi * d->Wi;
    if (S % 4 != 0) return false;                                // a 16-byte dY piece never straddles the end of a clip
    if ((long)d->N * d->Cin * S >= (1L << 29) || (long)d->N * d->Cout * S >= (1L << 29)) return false;   // int byte offsets
    const int M = d->Cout, p9 = (M + 143) / 144 * 144, p8 = (M + 127) / 128 * 128, pm = M <= 64 ? 64 : (p9 <= p8 ? p9 : p8);
    if (pm * 10 > M * 13) return false;                          // row padding above 30 %: the plain kernel's tiles fit better
    if ((long)d->N * S < 16384) return false;                    // too few voxels to give the workgroups useful slices
    if (x != nullptr && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return false;
    return true;
}

static size_t ww_align(size_t b) { return (b + 255) & ~(size_t)255; }

// F(4,3) form (conv_wgrad_wino4_kernel): rows in 144- or 128-row tiles, one 8-wave workgroup per CU
struct WgradWino4Plan {
    int tma, tmb, tiles_m, tiles_n, slices, chunks_per_slice, Cpad, Kp;
};

static bool wgrad_wino4_applicable(const zsv_conv_desc* d) {
    if (ZSV_KNOB(NO_WGRAD_WINO4)) return false;
    const long S = (long)d->Ti * d->Hi * d->Wi;
    return d->Wi % 4 == 0 && S % 32 == 0 && d->Cout > 64;
}

static WgradWino4Plan wgrad_wino4_plan(const zsv_conv_desc* d) {
    WgradWino4Plan pl;
    const int M = d->Cout;
    pl.Cpad = (d->Cin + 15) / 16 * 16;
    pl.Kp = 3 * d->kT * pl.Cpad;
    const int p9 = (M + 143) / 144 * 144, p8 = (M + 127) / 128 * 128;
    pl.tma = p9 <= p8 ? 5 : 4;
    pl.tmb = 4;
    const int bm = 16 * (pl.tma + pl.tmb);
    pl.tiles_m = (M + bm - 1) / bm;
    pl.tiles_n = (pl.Kp + 63) / 64;
    const long chunks = (long)d->N * ((long)d->Ti * d->Hi * d->Wi / 32);
    const long tiles = (long)pl.tiles_m * pl.tiles_n, resident = 256;       // one workgroup per CU
    const double voxels = (double)chunks * 32.0;
    const double t_mfma = 3.0 * (double)(pl.tiles_m * bm) * (double)(pl.tiles_n * 64) * voxels / 1.1e14;
    const double t_slice = 2.0 * 6.0 * (double)M * pl.Kp * sizeof(float) / 6.0e12;
    long max_sl = chunks / 24;
    if (max_sl < 1) max_sl = 1;
    if (max_sl > 4096) max_sl = 4096;
    long sl = 1;
    double best = 1e300;
    for (long c = 1; c <= max_sl; ++c) {
        const long wgs = tiles * c, rounds = (wgs + resident - 1) / resident;
        const double cost = t_mfma * (double)(rounds * resident) / (double)wgs + t_slice * (double)c;
        if (cost < best * 0.999) { best = cost; sl = c; }
    }
    if (const char* e = ZSV_KNOB(WGRAD_WINO_SLICES)) sl = atol(e);
    if (sl < 1) sl = 1;
    if (sl > chunks) sl = chunks;
    pl.chunks_per_slice = (int)((chunks + sl - 1) / sl);
    pl.slices = (int)((chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice);
    return pl;
}

size_t wgrad_wino_workspace_bytes(const zsv_conv_desc* d) {
    const WgradWinoPlan pl = wgrad_wino_plan(d);
    const size_t S = (size_t)d->Ti * d->Hi * d->Wi;
    size_t slabs = ww_align((size_t)pl.slices * 4 * d->Cout * pl.Kp * sizeof(float));
    if (d->Wi % 4 == 0 && S % 32 == 0 && d->Cout > 64) {           // (the knob may change between the query and the call)
        const WgradWino4Plan p4 = wgrad_wino4_plan(d);
        const size_t s4 = ww_align((size_t)p4.slices * 6 * d->Cout * p4.Kp * sizeof(float));
        if (s4 > slabs) slabs = s4;
    }
    return slabs + S * sizeof(unsigned);
}

template <int TMA, int TMB>
static int wgrad_wino4_launch(const WgradWinoParams& p, int slices, hipStream_t stream, const float* x, const float* dy,
                              const unsigned* vm, float* out) {
    constexpr int LDS_BYTES = 2 * (((16 * (TMA + TMB) * 9 + 63) / 64 + 11 + 1) * 1024);
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wgrad_wino4_kernel<TMA, TMB>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((conv_wgrad_wino4_kernel<TMA, TMB>), dim3((unsigned)(p.tiles_mn * slices)), dim3(512), LDS_BYTES, stream, p, x,
                       dy, vm, out);
    return launch_status();
}

template <int TM, int TN, bool EDGE>
static int wgrad_wino_launch(const WgradWinoParams& p, int slices, hipStream_t stream, const float* x, const float* dy,
                             const unsigned* vm, float* out) {
    constexpr int LDS_BYTES = 2 * (((16 * TM * 9 + 63) / 64 + 11 * TN + 1) * 1024);
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wgrad_wino_kernel<TM, TN, EDGE>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((conv_wgrad_wino_kernel<TM, TN, EDGE>), dim3((unsigned)(p.tiles_mn * slices)), dim3(256), LDS_BYTES, stream, p, x,
                       dy, vm, out);
    return launch_status();
}

int wgrad_wino(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
               hipStream_t stream, const unsigned* vm_ext) {
    const bool f43 = wgrad_wino4_applicable(d);
    WgradWinoPlan pl = wgrad_wino_plan(d);
    WgradWino4Plan p4{};
    if (f43) {
        p4 = wgrad_wino4_plan(d);
        pl.tiles_m = p4.tiles_m; pl.tiles_n = p4.tiles_n; pl.slices = p4.slices; pl.chunks_per_slice = p4.chunks_per_slice;
    }
    const int points = f43 ? 6 : 4;
    if (!workspace || workspace_bytes < wgrad_wino_workspace_bytes(d)) return ZSV_E_WORKSPACE;
    WgradWinoParams p;
    p.M = d->Cout; p.Cin = d->Cin; p.Cpad = pl.Cpad; p.R = 3 * d->kT; p.Kp = pl.Kp;
    p.S = d->Ti * d->Hi * d->Wi; p.HW = d->Hi * d->Wi; p.W = d->Wi; p.kT = d->kT;
    p.cpc = (p.S + 31) / 32;
    p.chunks_total = d->N * p.cpc;
    p.chunks_per_slice = pl.chunks_per_slice;
    p.x_bytes = 4u * (unsigned)((long)d->N * d->Cin * p.S);
    p.dy_bytes = 4u * (unsigned)((long)d->N * d->Cout * p.S);
    p.vm_bytes = 4u * (unsigned)p.S;
    p.tiles_m = pl.tiles_m; p.tiles_mn = pl.tiles_m * pl.tiles_n;
    float* slabs = (float*)workspace;
    // the tap-validity words depend on the geometry only: a caller that keeps them (zsv_conv3d_wgrad_masked) saves the launch
    const unsigned* vm = vm_ext;
    int st = ZSV_OK;
    if (vm == nullptr) {
        unsigned* own = (unsigned*)((char*)workspace + wgrad_wino_workspace_bytes(d) - (size_t)p.S * sizeof(unsigned));
        st = wgrad_vmask(d, own, stream);
        if (st) return st;
        vm = own;
    }
    const bool edge = p.S % 32 != 0 || d->Wi % 4 != 0;
    if (f43)
        st = p4.tma == 5 ? wgrad_wino4_launch<5, 4>(p, pl.slices, stream, x, dy, vm, slabs)
                         : wgrad_wino4_launch<4, 4>(p, pl.slices, stream, x, dy, vm, slabs);
    else if (pl.tm == 9)
        st = edge ? wgrad_wino_launch<9, 1, true>(p, pl.slices, stream, x, dy, vm, slabs)
                  : wgrad_wino_launch<9, 1, false>(p, pl.slices, stream, x, dy, vm, slabs);
    else if (pl.tm == 8)
        st = edge ? wgrad_wino_launch<8, 1, true>(p, pl.slices, stream, x, dy, vm, slabs)
                  : wgrad_wino_launch<8, 1, false>(p, pl.slices, stream, x, dy, vm, slabs);
    else
        st = edge ? wgrad_wino_launch<4, 2, true>(p, pl.slices, stream, x, dy, vm, slabs)
                  : wgrad_wino_launch<4, 2, false>(p, pl.slices, stream, x, dy, vm, slabs);
    if (st) return st;
    const long n = (long)d->Cout * p.R * pl.Cpad;
    long blocks = (n + 31) / 32;
    if (blocks > 8192) blocks = 8192;
    if (points == 6)
        hipLaunchKernelGGL(wgrad_wino_sum_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)slabs, dw, d->Cout,
                           d->Cin, p.R, pl.Cpad, pl.slices);
    else
        hipLaunchKernelGGL(wgrad_wino_sum_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)slabs, dw, d->Cout,
                           d->Cin, p.R, pl.Cpad, pl.slices);
    return launch_status();
}

}  // namespace zsv
