// wgrad_bf16.hip -- weight gradient of the stride-1 "same" convolutions on CHANNELS-LAST bf16 operands (the bf16 training step,
// amp.py; the reference's conv3d backward under autocast, main.py:172,195 with resnet.py:40-52 / 23-30).
//
//     dW[co][ci][a][b][c] = sum over voxels v of  dz[v][co] * x[v + shift(a,b,c)][ci]      (zero outside the clip)
//
// fp32 accumulation of bf16 x bf16 products on v_mfma_f32_16x16x32_bf16.  The contraction index is the VOXEL, which is the row
// index of both operands in memory ([voxel][channel], channels contiguous) -- the wrong way round for the matrix core, whose
// operands want 8 consecutive k per lane.  gfx950's transposed LDS read (ds_read_b64_tr_b16: a 4-row x 16-column block delivered
// column-major, cdna_hip_programming.md T10) turns the rows as they lie in memory into MFMA operands without a transpose pass:
// both operands are staged by LDS-DMA exactly as they are stored and read back transposed.
//
// Roles.  One operand is read at the voxel itself ("Q"), the other at the tap-shifted voxel ("P"); a wave owns 16 P channels,
// all TAPS = NIMG x KW shifts of them, and QB blocks of 16 Q channels: TAPS x QB accumulator tiles (27 of them in both shipped
// shapes), 2 transposed reads per operand fragment, TAPS + QB fragments per 32-voxel step for TAPS x QB MFMAs.
//   * kW = 3 (1x3x3, 3x3x3): P = x (input, Cin a multiple of 64 in the reference's trunks), Q = dz, QB = 3; NIMG = 3 row taps
//     (kh, or kt-by-kt groups for 3x3x3) each staged as ONE image of 32 + 2 rows that serves the three kw taps (a kw tap is a
//     row shift of the flattened voxel index); the w-border -- where the flattened shift wraps into the neighbouring row -- is a
//     per-voxel zeroing of the dz fragment in registers (two masked copies of each Q fragment);
//   * 3x1x1: P = dz read at v - shift (Cout a multiple of 64), Q = x, QB = 9, NIMG = 3 frame shifts, KW = 1, no masks at all:
//     a frame outside the clip is a zero image row.
// A workgroup = 4 waves = 64 P channels x (16 QB) Q channels x 3 row taps over a slice of the voxels; slices are summed in order
// by wgrad_cl_reduce_kernel (reproducible), which also writes torch's (Cout, Cin, kT, kH, kW) layout.
// LDS image: panels of [row][64 channels = 128 B], the 16-byte slots of a row XOR-swizzled by ((row>>1)&1)*2 ^ ((row>>3)&1)*4 on
// the DMA's source side, which makes every transposed read conflict-free for ANY row shift (the two 4-row runs of a 32-lane half
// alternate bank halves by row parity and differ in bit 1 / bit 3 of the row otherwise).
// Bound: L2 -> LDS operand traffic and LDS read bandwidth (24 transposed reads per 27 MFMAs), not the matrix pipe.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "conv_params.h"
#include "zsv_common.h"
#include "zsv_hip.h"
#include "knobs.h"

namespace zsv {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* wlds_ptr_t;

struct WgradClParams {
    int P;                      // rows of the Q operand = voxels the contraction runs over (N*To*Ho*Wo; stride 1: = input voxels)
    int T, H, W;                // extents those voxels are decoded in (the OUTPUT's; stride 1: = the input's)
    Magic mW, mH, mT;
    int ppitch, qpitch;         // channel pitches of the P / Q tensors (elements)
    unsigned p_bytes, q_bytes;  // sizes of the two tensors (buffer descriptors: a row outside reads as zeros)
    int pW;                     // (KW - 1) / 2
    int sgn;                    // +1: P = x read at v + shift; -1: P = dz read at v - shift
    int kH, pT, pH;             // row tap r = a * kH + b -> (a - pT, b - pH)
    // GATHER form (strided convolutions, 1x1x1): P = x, one image per TAP, read at (to*sT + a - pT, ho*sH + b - pH, wo*sW + c - pW)
    int kW3, sT, sH, sW, pWg;   // kernel width (tap -> (a, b, c)), strides, w padding
    int Ti, Hi, Wi;             // input extents
    int qgroups, ppanels, imggroups;
    int taps;                   // kT * kH * kW
    int chunks, chunks_per_slice;
    int Ptot, Qpad;             // partial layout [slice][tap][Ptot][Qpad]
};

__device__ __forceinline__ int row_sz(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }

__device__ __forceinline__ wu32x2 tr_read(unsigned addr) {
    wu32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void tie(wu32x2& v) { asm volatile("" : "+v"(v)); }

// GATHER: one image per tap, rows fetched at strided input coordinates (KW = 1; no flattened shifts, no border masks)
template <int NIMG, int KW, int QB, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void wgrad_cl_kernel(WgradClParams prm, const __bf16* __restrict__ Pt,
                                                          const __bf16* __restrict__ Qt, float* __restrict__ part) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!GATHER || KW == 1, "the gather form stages one image per tap");
    constexpr int TAPS = NIMG * KW;
    constexpr int PR8 = (32 + KW - 1 + 7) / 8;            // 1-KiB pieces (8 rows) per P image: 5 (KW = 3) or 4
    constexpr int QPAN = (QB * 16 + 63) / 64;             // Q panels of 64 channels
    constexpr int P_BYTES = NIMG * PR8 * 1024, Q_BYTES = QPAN * 4 * 1024;
    constexpr int STAGE = P_BYTES + Q_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (slice, image group, P panel, Q group): the blocks of one voxel slice read the same operand rows; consecutive logical
    // ids share an XCD (one L2) -- hardware deals consecutive blockIdx round-robin over the eight XCDs
    int b = xcd_tile(gridDim.x, blockIdx.x);
    const int qg = b % prm.qgroups; b /= prm.qgroups;
    const int pp = b % prm.ppanels; b /= prm.ppanels;
    const int ig = b % prm.imggroups;
    const int slice = b / prm.imggroups;
    const int c_begin = slice * prm.chunks_per_slice;
    int c_end = c_begin + prm.chunks_per_slice;
    if (c_end > prm.chunks) c_end = prm.chunks;

    // ---- DMA side: this lane's (row in piece, slot).  Rows come through buffer descriptors: one 32-bit byte offset per lane, a row
    // outside the clip (or a channel past the pitch) gets an offset past the descriptor's range and reads as zeros -- no pointer
    // selects, no zero line.
    const int drow = lane >> 3, dpos = lane & 7;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Pt), 0, prm.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Qt), 0, prm.q_bytes, 0x00020000);
    int da[NIMG], db[NIMG], dc[NIMG], dshift[NIMG];
#pragma unroll
    for (int i = 0; i < NIMG; ++i) {
        const int rt = ig * NIMG + i;
        if (GATHER) {
            const int c = rt % prm.kW3, ab = rt / prm.kW3;
            const int a = ab / prm.kH, bb = ab - a * prm.kH;
            da[i] = a - prm.pT; db[i] = bb - prm.pH; dc[i] = c - prm.pWg;
            dshift[i] = 0;
        } else {
            const int a = rt / prm.kH, bb = rt - a * prm.kH;
            da[i] = prm.sgn * (a - prm.pT);
            db[i] = prm.sgn * (bb - prm.pH);
            dc[i] = 0;
            dshift[i] = 2 * prm.ppitch * ((da[i] * prm.H + db[i]) * prm.W);      // bytes
        }
    }
    const int p_ch0 = pp * 64, q_ch0 = qg * QB * 16;
    const int p_row_bytes = 2 * prm.ppitch, q_row_bytes = 2 * prm.qpitch;

    auto issue = [&](int chunk, int stage) {
        const int v0 = chunk * 32;
        unsigned char* base = lds + stage * STAGE;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int rp = wave + 4 * k;
            if (rp < PR8) {
                const int r = 8 * rp + drow;
                const int u = v0 + r - prm.pW;
                const int slot = dpos ^ (row_sz(r) << 1);
                const int ch = p_ch0 + 8 * slot;
                const bool inside = (unsigned)u < (unsigned)prm.P && ch < prm.ppitch;
                const unsigned uu = inside ? (unsigned)u : 0u;
                const unsigned q1 = mdiv(uu, prm.mW);
                const unsigned q2 = mdiv(q1, prm.mH);
                const unsigned q3 = mdiv(q2, prm.mT);
                const int w = (int)(uu - q1 * prm.W), h = (int)(q1 - q2 * prm.H), t = (int)(q2 - q3 * prm.T);
                if (GATHER) {
                    const int ti0 = t * prm.sT, hi0 = h * prm.sH, wi0 = w * prm.sW;
                    const int nbase = (int)q3 * prm.Ti;
#pragma unroll
                    for (int i = 0; i < NIMG; ++i) {
                        const int ti = ti0 + da[i], hi = hi0 + db[i], wi = wi0 + dc[i];
                        const bool ok = inside && (unsigned)ti < (unsigned)prm.Ti && (unsigned)hi < (unsigned)prm.Hi && (unsigned)wi < (unsigned)prm.Wi;
                        const unsigned off = (unsigned)(((nbase + ti) * prm.Hi + hi) * prm.Wi + wi) * (unsigned)p_row_bytes + 2u * (unsigned)ch;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (wlds_ptr_t)(base + (i * PR8 + rp) * 1024), 16, (int)(ok ? off : OOB), 0, 0, 0);
                    }
                } else {
                    const unsigned off0 = uu * (unsigned)p_row_bytes + 2u * (unsigned)ch;
#pragma unroll
                    for (int i = 0; i < NIMG; ++i) {
                        const bool ok = inside && (unsigned)(t + da[i]) < (unsigned)prm.T && (unsigned)(h + db[i]) < (unsigned)prm.H;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (wlds_ptr_t)(base + (i * PR8 + rp) * 1024), 16,
                                                                 (int)(ok ? off0 + (unsigned)dshift[i] : OOB), 0, 0, 0);
                    }
                }
            }
        }
        {
            const int r = 8 * wave + drow;
            const int slot = dpos ^ (row_sz(r) << 1);
            // (a row past the last voxel lies past the descriptor's range by itself; a channel past the pitch must be sent there)
            const unsigned off0 = (unsigned)(v0 + r) * (unsigned)q_row_bytes + 2u * (unsigned)(q_ch0 + 8 * slot);
#pragma unroll
            for (int pn = 0; pn < QPAN; ++pn) {
                const bool ok = q_ch0 + pn * 64 + 8 * slot < prm.qpitch && v0 + r < prm.P;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (wlds_ptr_t)(base + P_BYTES + (pn * 4 + wave) * 1024), 16,
                                                         (int)(ok ? off0 + 128u * (unsigned)pn : OOB), 0, 0, 0);
            }
        }
    };

    // ---- fragment side -------------------------------------------------------------------------------------------------------
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int rb = 8 * g + q4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // P fragment of tap (image i, kw c), k half hh: image row rb + 4 hh + c, channel block `wave` of the panel
    unsigned p_off[KW][2];
#pragma unroll
    for (int c = 0; c < KW; ++c)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int r = rb + 4 * hh + c;
            p_off[c][hh] = (r >> 3) * 1024 + (r & 7) * 128 + ((((wave ^ row_sz(r)) << 1) | (p4 >> 1)) << 4) + ((p4 & 1) << 3);
        }
    // Q fragment of block j: rows rb + 4 hh, channel block j & 3 of panel j >> 2
    unsigned q_off[QB][2];
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int r = rb + 4 * hh;
            q_off[j][hh] = P_BYTES + (j >> 2) * 4096 + (r >> 3) * 1024 + (r & 7) * 128 +
                           (((((j & 3) ^ row_sz(r)) << 1) | (p4 >> 1)) << 4) + ((p4 & 1) << 3);
        }

    f32x4 acc[TAPS][QB];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int j = 0; j < QB; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (c_begin < c_end) issue(c_begin, 0);
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const int stage = (chunk - c_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (chunk + 1 < c_end) issue(chunk + 1, stage ^ 1);
        const unsigned sb = lds_base + stage * STAGE;

        wu32x2 qlo[QB], qhi[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            qlo[j] = tr_read(sb + q_off[j][0]);
            qhi[j] = tr_read(sb + q_off[j][1]);
        }
        // w-border masks of this lane's 8 voxels (kw = 0 needs w >= 1, kw = 2 needs w <= W - 2); voxel j sits in half (j & 1) of
        // dword (j >> 1) of the fragment.  With W >= 8 at most one voxel of the eight has w == 0 and at most one w == W - 1.
        unsigned m0[4], m2[4];
        if (KW == 3) {
            const int vb = chunk * 32 + 8 * g;
            int w = (int)((unsigned)vb - mdiv((unsigned)vb, prm.mW) * prm.W);
            if (prm.W >= 8) {
                const int j0 = w == 0 ? 0 : prm.W - w;                 // first voxel with w == 0 (>= 8: none)
                const int j2 = prm.W - 1 - w;                          //                  w == W - 1
                const unsigned k0 = (j0 & 1) ? 0x0000ffffu : 0xffff0000u, k2 = (j2 & 1) ? 0x0000ffffu : 0xffff0000u;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    m0[d] = (j0 >> 1) == d ? k0 : 0xffffffffu;
                    m2[d] = (j2 >> 1) == d ? k2 : 0xffffffffu;
                }
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) { m0[d] = 0xffffffffu; m2[d] = 0xffffffffu; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned keep = (j & 1) ? 0x0000ffffu : 0xffff0000u;
                    if (w == 0) m0[j >> 1] &= keep;
                    if (w == prm.W - 1) m2[j >> 1] &= keep;
                    w = (w + 1 == prm.W) ? 0 : w + 1;
                }
            }
        }
        wbf16x8 qf[KW][QB];                                            // the Q fragments of the three kw taps: formed once per chunk
        bool q_ready = false;
#pragma unroll
        for (int i = 0; i < NIMG; ++i) {
            wu32x2 plo[KW], phi[KW];
#pragma unroll
            for (int c = 0; c < KW; ++c) {
                plo[c] = tr_read(sb + i * PR8 * 1024 + p_off[c][0]);
                phi[c] = tr_read(sb + i * PR8 * 1024 + p_off[c][1]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < KW; ++c) { tie(plo[c]); tie(phi[c]); }
            if (!q_ready) {
                q_ready = true;
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    tie(qlo[j]);
                    tie(qhi[j]);
                    const wu32x4 qv = {qlo[j][0], qlo[j][1], qhi[j][0], qhi[j][1]};
                    if (KW == 3) {
                        const wu32x4 q0 = {qv[0] & m0[0], qv[1] & m0[1], qv[2] & m0[2], qv[3] & m0[3]};
                        const wu32x4 q2 = {qv[0] & m2[0], qv[1] & m2[1], qv[2] & m2[2], qv[3] & m2[3]};
                        qf[0][j] = __builtin_bit_cast(wbf16x8, q0);
                        qf[KW / 2][j] = __builtin_bit_cast(wbf16x8, qv);
                        qf[KW - 1][j] = __builtin_bit_cast(wbf16x8, q2);
                    } else {
                        qf[0][j] = __builtin_bit_cast(wbf16x8, qv);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < KW; ++c) {
                const wu32x4 pv = {plo[c][0], plo[c][1], phi[c][0], phi[c][1]};
                const wbf16x8 pf = __builtin_bit_cast(wbf16x8, pv);
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    acc[i * KW + c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, qf[c][j], acc[i * KW + c][j], 0, 0, 0);
            }
        }
    }

    // ---- partial sums: part[slice][tap][P channel][Q channel] ----------------------------------------------------------------
    const int pch = pp * 64 + wave * 16 + 4 * g;
    const int qch = qg * QB * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < NIMG; ++i)
#pragma unroll
        for (int c = 0; c < KW; ++c) {
            const int tap = (ig * NIMG + i) * KW + c;
            float* out = part + ((size_t)(slice * prm.taps + tap) * prm.Ptot + pch) * prm.Qpad + qch;
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) out[(size_t)e * prm.Qpad + 16 * j] = acc[i * KW + c][j][e];
        }
#endif
}

// dW (Cout, Cin, taps) = sum over slices.  p_is_cin: the P operand was x (rows of `part` are input channels).
// Block = 32 outputs x 8 slice lanes: lane sl adds slices sl, sl + 8, ... (four loads in flight), the 8 lanes are added in lane
// order through LDS: a fixed order, so the result is reproducible.  (One thread walking all 256 slices of an output serially
// took 0.17 ms per call: latency, not bytes.)
__global__ __launch_bounds__(256) void wgrad_cl_reduce_kernel(const float* __restrict__ part, int slices, int taps, int Ptot, int Qpad,
                                                              int Pch, int Qch, int p_is_cin, float* __restrict__ dw) {
    __shared__ float red[8][33];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long idx = (long)blockIdx.x * 32 + o;
    const long total = (long)taps * Pch * Qch;
    const bool live = idx < total;
    int q = 0, p = 0, tap = 0;
    if (live) {
        q = (int)(idx % Qch);
        const long r = idx / Qch;
        p = (int)(r % Pch);
        tap = (int)(r / Pch);
    }
    const size_t stride = (size_t)taps * Ptot * Qpad;
    const float* src = part + ((size_t)tap * Ptot + p) * Qpad + q;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (live) {
        int k = sl;
        for (; k + 24 < slices; k += 32) {
            s0 += src[(size_t)k * stride];
            s1 += src[(size_t)(k + 8) * stride];
            s2 += src[(size_t)(k + 16) * stride];
            s3 += src[(size_t)(k + 24) * stride];
        }
        for (; k < slices; k += 8) s0 += src[(size_t)k * stride];
    }
    red[sl][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && live) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) s += red[l][o];
        const int co = p_is_cin ? q : p, ci = p_is_cin ? p : q;
        const int Cin = p_is_cin ? Pch : Qch;
        dw[((size_t)co * Cin + ci) * taps + tap] = s;
    }
}

struct WgradClPlan {
    bool ok;
    // 0: stride 1 "same", kW = 3: P = x, Q = dz, QB = 3 (three kw taps per row-tap image);  1: stride 1 "same" 3x1x1: P = dz, Q = x, QB = 9
    // gather forms (strided convolutions, 1x1x1; one image per tap, P = x, Q = dz):  2: kH = kW = 3 (9 taps per workgroup, QB = 3);
    // 3: 3x1x1 (QB = 9);  4: 1x1x1 (QB = 9)
    int mode;
    int qb, qgroups, ppanels, imggroups, taps, chunks, chunks_per_slice, slices, Ptot, Qpad, Pch, Qch, ppitch, qpitch;
    long rows;                 // voxels of the contraction
};

static inline int rup(int a, int b) { return (a + b - 1) / b * b; }

static WgradClPlan wgrad_cl_plan(const zsv_conv_desc* d) {
    WgradClPlan pl{};
    pl.ok = false;
    if (d == nullptr || conv_check(d) != ZSV_OK) return pl;
    if (ZSV_KNOB(BF16_NO_WGRAD) != nullptr) return pl;
    if (d->Cin <= 4) return pl;
    const int cinp = rup(d->Cin, 32), coutp = rup(d->Cout, 32);
    const long in_vox = (long)d->N * d->Ti * d->Hi * d->Wi, out_vox = (long)d->N * d->To * d->Ho * d->Wo;
    // 32-bit byte offsets inside the two buffer descriptors
    if (in_vox * cinp * 2 >= (1L << 32) - 64 || out_vox * coutp * 2 >= (1L << 32) - 64 || out_vox >= (1L << 30)) return pl;
    const bool same = d->sT == 1 && d->sH == 1 && d->sW == 1 && d->To == d->Ti && d->Ho == d->Hi && d->Wo == d->Wi &&
                      2 * d->pT == d->kT - 1 && 2 * d->pH == d->kH - 1 && 2 * d->pW == d->kW - 1;
    const int rowtaps = d->kT * d->kH;
    pl.Pch = d->Cin; pl.Qch = d->Cout; pl.ppitch = cinp; pl.qpitch = coutp;
    pl.imggroups = 1;
    if (same && d->kW == 3 && (rowtaps == 3 || rowtaps == 9) && (d->kH == 3 || d->kH == 1) && (d->kT == 3 || d->kT == 1)) {
        pl.mode = 0; pl.qb = 3;
        pl.imggroups = rowtaps / 3;
    } else if (same && d->kW == 1 && d->kH == 1 && d->kT == 3) {
        pl.mode = 1; pl.qb = 9;
        pl.Pch = d->Cout; pl.Qch = d->Cin; pl.ppitch = coutp; pl.qpitch = cinp;
    } else if (ZSV_KNOB(BF16_NO_WGRAD_GATHER) != nullptr) {
        return pl;
    } else if (d->kH == 3 && d->kW == 3 && (d->kT == 1 || d->kT == 3)) {
        pl.mode = 2; pl.qb = 3;
        pl.imggroups = d->kT;
    } else if (d->kT == 3 && d->kH == 1 && d->kW == 1) {
        pl.mode = 3; pl.qb = 9;
    } else if (d->kT == 1 && d->kH == 1 && d->kW == 1) {
        pl.mode = 4; pl.qb = 9;
    } else {
        return pl;
    }
    pl.rows = out_vox;
    pl.taps = d->kT * d->kH * d->kW;
    pl.ppanels = (pl.Pch + 63) / 64;
    pl.Ptot = pl.ppanels * 64;
    const int qblocks = (pl.Qch + 15) / 16;
    pl.qgroups = (qblocks + pl.qb - 1) / pl.qb;
    pl.Qpad = pl.qgroups * pl.qb * 16;
    pl.chunks = (int)((out_vox + 31) / 32);
    const int base = pl.ppanels * pl.qgroups * pl.imggroups;
    const char* e = ZSV_KNOB(BF16_WGRAD_WGS);
    const int target = e ? atoi(e) : 384;          // workgroups per launch (measured 256 / 384 / 512 / 768: 384 gave the best step); every slice costs a partial-sum round trip
    int slices = (target + base - 1) / base;
    if (slices > pl.chunks) slices = pl.chunks;
    if (slices < 1) slices = 1;
    pl.chunks_per_slice = (pl.chunks + slices - 1) / slices;
    pl.slices = (pl.chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice;
    pl.ok = true;
    return pl;
}

template <int NIMG, int KW, int QB, bool GATHER>
static int wgrad_cl_launch(const WgradClParams& p, int blocks, hipStream_t stream, const __bf16* Pt, const __bf16* Qt, float* part) {
    constexpr int PR8 = (32 + KW - 1 + 7) / 8, QPAN = (QB * 16 + 63) / 64;
    constexpr int LDS_BYTES = 2 * (NIMG * PR8 + QPAN * 4) * 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)wgrad_cl_kernel<NIMG, KW, QB, GATHER>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((wgrad_cl_kernel<NIMG, KW, QB, GATHER>), dim3((unsigned)blocks), dim3(256), LDS_BYTES, stream, p, Pt, Qt, part);
    return launch_status();
}

}  // namespace zsv

using namespace zsv;

extern "C" {

size_t zsv_conv3d_bf16_wgrad_workspace_bytes(const zsv_conv_desc* d) {
    const WgradClPlan pl = wgrad_cl_plan(d);
    if (!pl.ok) return 0;
    return (size_t)pl.slices * pl.taps * pl.Ptot * pl.Qpad * sizeof(float);
}

int zsv_conv3d_bf16_wgrad(const zsv_conv_desc* d, const void* x, const void* dz, float* dw, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (d == nullptr) return ZSV_E_NULL;
    const WgradClPlan pl = wgrad_cl_plan(d);
    if (!pl.ok) return ZSV_E_UNSUPPORTED;
    if (!x || !dz || !dw || !workspace) return ZSV_E_NULL;
    if (workspace_bytes < (size_t)pl.slices * pl.taps * pl.Ptot * pl.Qpad * sizeof(float)) return ZSV_E_WORKSPACE;
    WgradClParams p;
    const bool gather = pl.mode >= 2;
    p.P = (int)pl.rows;
    p.T = gather ? d->To : d->Ti; p.H = gather ? d->Ho : d->Hi; p.W = gather ? d->Wo : d->Wi;
    p.mW = make_magic((unsigned)p.W); p.mH = make_magic((unsigned)p.H); p.mT = make_magic((unsigned)p.T);
    p.ppitch = pl.ppitch; p.qpitch = pl.qpitch;
    const long in_vox = (long)d->N * d->Ti * d->Hi * d->Wi;
    const unsigned x_bytes = (unsigned)(in_vox * rup(d->Cin, 32) * 2), dz_bytes = (unsigned)(pl.rows * rup(d->Cout, 32) * 2);
    p.p_bytes = pl.mode == 1 ? dz_bytes : x_bytes;
    p.q_bytes = pl.mode == 1 ? x_bytes : dz_bytes;
    p.pW = gather ? 0 : d->pW;
    p.sgn = pl.mode == 1 ? -1 : 1;
    p.kH = d->kH; p.pT = d->pT; p.pH = d->pH;
    p.kW3 = d->kW; p.sT = d->sT; p.sH = d->sH; p.sW = d->sW; p.pWg = d->pW;
    p.Ti = d->Ti; p.Hi = d->Hi; p.Wi = d->Wi;
    p.qgroups = pl.qgroups; p.ppanels = pl.ppanels; p.imggroups = pl.imggroups;
    p.taps = pl.taps;
    p.chunks = pl.chunks; p.chunks_per_slice = pl.chunks_per_slice;
    p.Ptot = pl.Ptot; p.Qpad = pl.Qpad;
    const int blocks = pl.slices * pl.imggroups * pl.ppanels * pl.qgroups;
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    const __bf16 *xb = (const __bf16*)x, *zb = (const __bf16*)dz;
    int st;
    switch (pl.mode) {
        case 0: st = wgrad_cl_launch<3, 3, 3, false>(p, blocks, s, xb, zb, part); break;
        case 1: st = wgrad_cl_launch<3, 1, 9, false>(p, blocks, s, zb, xb, part); break;
        case 2: st = wgrad_cl_launch<9, 1, 3, true>(p, blocks, s, xb, zb, part); break;
        case 3: st = wgrad_cl_launch<3, 1, 9, true>(p, blocks, s, xb, zb, part); break;
        default: st = wgrad_cl_launch<1, 1, 9, true>(p, blocks, s, xb, zb, part); break;
    }
    if (st != ZSV_OK) return st;
    const long total = (long)pl.taps * pl.Pch * pl.Qch;
    hipLaunchKernelGGL(wgrad_cl_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, part, pl.slices, pl.taps, pl.Ptot,
                       pl.Qpad, pl.Pch, pl.Qch, pl.mode == 1 ? 0 : 1, dw);
    return launch_status();
}

}  // extern "C"
