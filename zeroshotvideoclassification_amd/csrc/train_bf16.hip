// train_bf16.hip -- the element-wise / reduction half of the bf16 TRAINING step (the reference's mixed-precision step,
// main.py:172 `with autocast():` + main.py:137,195-203 GradScaler): train-mode BatchNorm3d forward and backward, ReLU and the
// residual add on CHANNELS-LAST bf16 activations [R = N*T*H*W rows][Cp], Cp = channels rounded up to 32 (the layout of
// conv_bf16.hip), plus the layout converters that hand a tensor to the fp32 NCDHW kernels.
//
// What each kernel replaces (resnet.py:40-52,94-98,110-111 under autocast): aten::batch_norm on a bf16 input keeps fp32
// statistics / affine parameters and writes bf16; `out += residual; relu` are bf16 element-wise passes.  Here:
//     forward   stats (sum z, sum z^2 per channel: fp32 per thread, fixed-order fp32 per block, fp64 across blocks)
//               -> finalize (mean, invstd, scale a = gamma*invstd, shift b = beta - mean*a, running statistics, momentum as torch)
//               -> apply   y = relu?( fma(z, a, b) (+ res) )                                      one rounding to bf16
//     backward  reduce   g = dy * (y > 0)?;  sum g, sum g*z per channel
//               -> finalize dbeta = sum g, dgamma = invstd*(sum g*z - mean*sum g), dz = A*g + B*z + C  (A = gamma*invstd,
//                  B = -A*invstd*dgamma/R, C = -A*sum g/R - B*mean)
//               -> apply   dz (bf16), and the masked gradient g itself when a residual branch needs it
// Thread layout (all four passes): a row is G = Cp/8 sixteen-byte items; a 256-thread block is RL = 256/G "row lanes" x G items,
// thread (rl, o) keeps ONE channel octet for its whole life (its 8 coefficients live in registers) and walks rows rl, rl+RL, ...
// of the block's row range: consecutive threads read consecutive 16-byte items, consecutive row lanes consecutive rows.
// Bound: HBM (2-byte activations: half the bytes of the fp32 passes of batchnorm.hip).
#include <hip/hip_runtime.h>

#include "zsv_common.h"
#include "zsv_hip.h"

namespace zsv {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// round-to-nearest-even fp32 -> bf16 (finite inputs; NaN keeps a quiet NaN)
__device__ __forceinline__ unsigned to_bf16_bits(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return to_bf16_bits(lo) | (to_bf16_bits(hi) << 16); }

__device__ __forceinline__ void unpack8(const u32x4v r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf_lo(r[i]); v[2 * i + 1] = bf_hi(r[i]); }
}
__device__ __forceinline__ u32x4v pack8(const float (&v)[8]) {
    u32x4v r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack_bf16(v[2 * i], v[2 * i + 1]);
    return r;
}

struct ClShape {
    long R;              // rows
    int Cp, G, RL;       // channel pitch, 16-byte items per row, row lanes per block
    long rows_per_block;
};

// MODE 0: sums of (z, z*z).  MODE 1: sums of (g, g*z) with g = dy, MODE 2: the same with g = dy * (y > 0), MODE 3: the same
// with the mask recomputed from z and the forward's coefficients (fma(z, a, b) > 0: what the forward rounded to y -- a positive
// fp32 never rounds to a bf16 zero -- so the saved output need not be read: 4 of the backward's 14 bytes per element).
template <int MODE>
__global__ __launch_bounds__(256) void bn_cl_reduce_kernel(ClShape sh, const u32x4v* __restrict__ z, const u32x4v* __restrict__ dy,
                                                           const u32x4v* __restrict__ y, const float* __restrict__ fwd_coef,
                                                           float* __restrict__ partial) {
    extern __shared__ float red[];                 // [RL][2][Cp]
    const int o = threadIdx.x % sh.G, rl = threadIdx.x / sh.G;
    float s0[8], s1[8], fa[8], fb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
    if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { fa[j] = fwd_coef[o * 8 + j]; fb[j] = fwd_coef[sh.Cp + o * 8 + j]; }
    }
    const long row0 = (long)blockIdx.x * sh.rows_per_block;
    long row_end = row0 + sh.rows_per_block;
    if (row_end > sh.R) row_end = sh.R;
    if (rl < sh.RL) {
        for (long r = row0 + rl; r < row_end; r += sh.RL) {
            const long idx = r * sh.G + o;
            float zv[8];
            unpack8(__builtin_nontemporal_load(z + idx), zv);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s0[j] += zv[j]; s1[j] = fmaf(zv[j], zv[j], s1[j]); }
            } else {
                float gv[8];
                unpack8(__builtin_nontemporal_load(dy + idx), gv);
                if (MODE == 2) {
                    float yv[8];
                    unpack8(__builtin_nontemporal_load(y + idx), yv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
                }
                if (MODE == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) gv[j] = fmaf(zv[j], fa[j], fb[j]) > 0.f ? gv[j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { s0[j] += gv[j]; s1[j] = fmaf(gv[j], zv[j], s1[j]); }
            }
        }
        float* mine = red + (size_t)rl * 2 * sh.Cp + o * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { mine[j] = s0[j]; mine[sh.Cp + j] = s1[j]; }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * 2 * sh.Cp;
    for (int c = threadIdx.x; c < 2 * sh.Cp; c += 256) {
        float acc = 0.f;
        for (int l = 0; l < sh.RL; ++l) acc += red[(size_t)l * 2 * sh.Cp + c];     // fixed order: reproducible
        out[c] = acc;
    }
}

// Sum of the per-block partials of 16 channels: block = 16 channels x 16 slices of the nb blocks; each thread adds its slice in
// double (fixed order), the 16 slices are added in order by the first 16 threads.  (One thread per channel walking all ~2000
// partials serially took 0.32 ms per BatchNorm: 23 ms of a training step.)
__device__ __forceinline__ void partial_sums_16(const float* __restrict__ partial, int nb, int Cp, int c, double& s0, double& s1) {
    __shared__ double red[2][16][17];
    const int slice = threadIdx.x >> 4, cl = threadIdx.x & 15;
    double a = 0.0, b = 0.0;
    if (c < Cp) {
#pragma unroll 8
        for (int blk = slice; blk < nb; blk += 16) {
            a += (double)partial[(size_t)blk * 2 * Cp + c];
            b += (double)partial[(size_t)blk * 2 * Cp + Cp + c];
        }
    }
    red[0][slice][cl] = a;
    red[1][slice][cl] = b;
    __syncthreads();
    s0 = s1 = 0.0;
    if (slice == 0) {
        for (int l = 0; l < 16; ++l) { s0 += red[0][l][cl]; s1 += red[1][l][cl]; }
    }
}

// forward finalize: 16 (padded) channels per 256-thread block
__global__ __launch_bounds__(256) void bn_cl_fwd_finalize_kernel(const float* __restrict__ partial, int nb, int C, int Cp, long R,
                                          const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* __restrict__ running_mean,
                                          float* __restrict__ running_var, float momentum, float eps, float* __restrict__ save_mean,
                                          float* __restrict__ save_invstd, float* __restrict__ coef /*[2][Cp]*/) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double s, q;
    partial_sums_16(partial, nb, Cp, c, s, q);
    if ((threadIdx.x >> 4) != 0 || c >= Cp) return;
    if (c >= C) { coef[c] = 0.f; coef[Cp + c] = 0.f; return; }      // pad channels stay zero
    const double mean = s / (double)R;
    double var = q / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
    const double a = g * invstd;
    coef[c] = (float)a;
    coef[Cp + c] = (float)(bt - mean * a);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
    if (running_var) {
        const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// y = relu?(fma(z, a, b) (+ res))
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_cl_apply_kernel(ClShape sh, const u32x4v* __restrict__ z, const u32x4v* __restrict__ res,
                                                          const float* __restrict__ coef, u32x4v* __restrict__ y) {
    const int o = threadIdx.x % sh.G, rl = threadIdx.x / sh.G;
    if (rl >= sh.RL) return;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = coef[o * 8 + j]; b[j] = coef[sh.Cp + o * 8 + j]; }
    const long row0 = (long)blockIdx.x * sh.rows_per_block;
    long row_end = row0 + sh.rows_per_block;
    if (row_end > sh.R) row_end = sh.R;
    for (long r = row0 + rl; r < row_end; r += sh.RL) {
        const long idx = r * sh.G + o;
        float v[8];
        unpack8(__builtin_nontemporal_load(z + idx), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a[j], b[j]);
        if (RES) {
            float rv[8];
            unpack8(__builtin_nontemporal_load(res + idx), rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
        if (RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        y[idx] = pack8(v);
    }
}

// backward finalize: dgamma, dbeta and the three coefficients of dz = A*g + B*z + C
__global__ __launch_bounds__(256) void bn_cl_bwd_finalize_kernel(const float* __restrict__ partial, int nb, int C, int Cp, long R,
                                          const float* __restrict__ gamma,
                                          const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef /*[3][Cp]*/) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double sg, sgz;
    partial_sums_16(partial, nb, Cp, c, sg, sgz);
    if ((threadIdx.x >> 4) != 0 || c >= Cp) return;
    if (c >= C) { coef[c] = 0.f; coef[Cp + c] = 0.f; coef[2 * Cp + c] = 0.f; return; }
    const double mean = save_mean[c], invstd = save_invstd[c];
    const double dg = (sgz - mean * sg) * invstd;
    const double A = (gamma ? (double)gamma[c] : 1.0) * invstd;
    const double B = -A * invstd * dg / (double)R;
    const double Cc = -A * sg / (double)R - B * mean;
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)sg;
    coef[c] = (float)A;
    coef[Cp + c] = (float)B;
    coef[2 * Cp + c] = (float)Cc;
}

// dz = A*g + B*z + C with g = dy (* mask); MASK 0: none, 1: y > 0 (saved output), 2: fma(z, a, b) > 0 (forward coefficients);
// GOUT: also store g (the residual branch's gradient)
template <int MASK, bool GOUT>
__global__ __launch_bounds__(256) void bn_cl_bwd_apply_kernel(ClShape sh, const u32x4v* __restrict__ dy, const u32x4v* __restrict__ y,
                                                              const u32x4v* __restrict__ z, const float* __restrict__ coef,
                                                              const float* __restrict__ fwd_coef, u32x4v* __restrict__ dz,
                                                              u32x4v* __restrict__ gout) {
    const int o = threadIdx.x % sh.G, rl = threadIdx.x / sh.G;
    if (rl >= sh.RL) return;
    float A[8], B[8], Cc[8], fa[8], fb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { A[j] = coef[o * 8 + j]; B[j] = coef[sh.Cp + o * 8 + j]; Cc[j] = coef[2 * sh.Cp + o * 8 + j]; }
    if (MASK == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { fa[j] = fwd_coef[o * 8 + j]; fb[j] = fwd_coef[sh.Cp + o * 8 + j]; }
    }
    const long row0 = (long)blockIdx.x * sh.rows_per_block;
    long row_end = row0 + sh.rows_per_block;
    if (row_end > sh.R) row_end = sh.R;
    for (long r = row0 + rl; r < row_end; r += sh.RL) {
        const long idx = r * sh.G + o;
        float gv[8], zv[8];
        unpack8(__builtin_nontemporal_load(dy + idx), gv);
        unpack8(__builtin_nontemporal_load(z + idx), zv);
        if (MASK == 1) {
            float yv[8];
            unpack8(__builtin_nontemporal_load(y + idx), yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
        }
        if (MASK == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = fmaf(zv[j], fa[j], fb[j]) > 0.f ? gv[j] : 0.f;
        }
        if (GOUT) gout[idx] = pack8(gv);
        float dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = fmaf(A[j], gv[j], fmaf(B[j], zv[j], Cc[j]));
        dz[idx] = pack8(dv);
    }
}

// [N][S][Cp] bf16 -> (N, C, S) fp32: 64 voxels x 32 channels per block through LDS
__global__ __launch_bounds__(256) void cl_bf16_to_ncs_f32_kernel(const unsigned short* __restrict__ x, int S, int C, int Cp,
                                                                 float* __restrict__ out) {
    __shared__ float tile[64][33];
    const int s0 = blockIdx.x * 64, c0 = blockIdx.y * 32, n = blockIdx.z;
    {
        const int v = threadIdx.x >> 2, part = threadIdx.x & 3;           // 4 threads x 16 bytes = the 32 channels of a voxel
        const int s = s0 + v;
        float vals[8];
        if (s < S) {
            const u32x4v r = *(const u32x4v*)(x + ((size_t)n * S + s) * Cp + c0 + part * 8);
            unpack8(r, vals);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) vals[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[v][part * 8 + j] = vals[j];
    }
    __syncthreads();
    const int v = threadIdx.x & 63, cg = threadIdx.x >> 6;
    if (s0 + v < S) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + cg * 8 + i;
            if (c < C) out[((size_t)n * C + c) * S + s0 + v] = tile[v][cg * 8 + i];
        }
    }
}

// (N, C, S) fp32 -> [N][S][Cp] bf16 (pad channels zero)
__global__ __launch_bounds__(256) void ncs_f32_to_cl_bf16_kernel(const float* __restrict__ x, int S, int C, int Cp,
                                                                 unsigned short* __restrict__ out) {
    __shared__ float tile[32][65];
    const int s0 = blockIdx.x * 64, c0 = blockIdx.y * 32, n = blockIdx.z;
    {
        const int v = threadIdx.x & 63, cg = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + cg * 8 + i;
            tile[cg * 8 + i][v] = (c < C && s0 + v < S) ? x[((size_t)n * C + c) * S + s0 + v] : 0.f;
        }
    }
    __syncthreads();
    const int v = threadIdx.x >> 2, part = threadIdx.x & 3;
    if (s0 + v < S) {
        float vals[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[j] = tile[part * 8 + j][v];
        *(u32x4v*)(out + ((size_t)n * S + s0 + v) * Cp + c0 + part * 8) = pack8(vals);
    }
}

// gradient of the mean over the S voxels: dx[n][s][c] = dpooled[n][c] / S (bf16, pad channels zero)
__global__ void meanpool_cl_bwd_kernel(const float* __restrict__ dp, int S, int C, int Cp, long total, unsigned short* __restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cp);
    const long ns = i / Cp;
    const long n = ns / S;
    dx[i] = c < C ? (unsigned short)to_bf16_bits(dp[n * C + c] / (float)S) : (unsigned short)0;
}


// ---- C3D's training step in bf16 (network.py:147-163 under autocast): max-pool backward and ReLU mask + bias gradient ---------
// MaxPool3d (kernel == stride) backward on channels-last bf16: one thread per (window, 8-channel octet) finds the window's FIRST
// maximum in (t, h, w) scan order per channel -- aten::max_pool3d_with_indices' argmax -- and writes dy there and zero to the
// window's other voxels (a voxel belongs to exactly one window; voxels no window covers are zeroed by the caller).
__global__ void maxpool3d_cl_bwd_kernel(const u32x4v* __restrict__ dy, const u32x4v* __restrict__ x, int Ti, int Hi, int Wi, int G, int kT,
                                        int kH, int kW, int pT, int pH, int pW, int To, int Ho, int Wo, long total, u32x4v* __restrict__ dx) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int o = (int)(idx % G);
    long r = idx / G;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int to = (int)(r % To);
    const long n = r / To;
    float best[8], g[8];
    int arg[8];
    unpack8(dy[idx], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = -1; }
    int tap = 0;
    for (int a = 0; a < kT; ++a)
        for (int b = 0; b < kH; ++b)
            for (int c = 0; c < kW; ++c, ++tap) {
                const int t = to * kT + a - pT, h = ho * kH + b - pH, w = wo * kW + c - pW;
                if ((unsigned)t >= (unsigned)Ti || (unsigned)h >= (unsigned)Hi || (unsigned)w >= (unsigned)Wi) continue;
                float v[8];
                unpack8(x[(((n * Ti + t) * Hi + h) * (long)Wi + w) * G + o], v);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (v[j] > best[j] || arg[j] < 0) { best[j] = v[j]; arg[j] = tap; }
            }
    tap = 0;
    for (int a = 0; a < kT; ++a)
        for (int b = 0; b < kH; ++b)
            for (int c = 0; c < kW; ++c, ++tap) {
                const int t = to * kT + a - pT, h = ho * kH + b - pH, w = wo * kW + c - pW;
                if ((unsigned)t >= (unsigned)Ti || (unsigned)h >= (unsigned)Hi || (unsigned)w >= (unsigned)Wi) continue;
                float outv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) outv[j] = arg[j] == tap ? g[j] : 0.f;
                dx[(((n * Ti + t) * Hi + h) * (long)Wi + w) * G + o] = pack8(outv);
            }
}

// g = dy * (y > 0) (bf16) and the per-block partial sums of g per channel (the bias gradient): `relu(conv(x) + bias)` backward
__global__ __launch_bounds__(256) void relu_bias_bwd_cl_kernel(ClShape sh, const u32x4v* __restrict__ dy, const u32x4v* __restrict__ y,
                                                               u32x4v* __restrict__ gout, float* __restrict__ partial) {
    extern __shared__ float red[];                 // [RL][2][Cp] (second half unused: the layout of partial_sums_16)
    const int o = threadIdx.x % sh.G, rl = threadIdx.x / sh.G;
    float s0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = 0.f;
    const long row0 = (long)blockIdx.x * sh.rows_per_block;
    long row_end = row0 + sh.rows_per_block;
    if (row_end > sh.R) row_end = sh.R;
    if (rl < sh.RL) {
        for (long r = row0 + rl; r < row_end; r += sh.RL) {
            const long idx = r * sh.G + o;
            float gv[8], yv[8];
            unpack8(__builtin_nontemporal_load(dy + idx), gv);
            unpack8(__builtin_nontemporal_load(y + idx), yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { gv[j] = yv[j] > 0.f ? gv[j] : 0.f; s0[j] += gv[j]; }
            gout[idx] = pack8(gv);
        }
        float* mine = red + (size_t)rl * 2 * sh.Cp + o * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { mine[j] = s0[j]; mine[sh.Cp + j] = 0.f; }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * 2 * sh.Cp;
    for (int c = threadIdx.x; c < 2 * sh.Cp; c += 256) {
        float acc = 0.f;
        for (int l = 0; l < sh.RL; ++l) acc += red[(size_t)l * 2 * sh.Cp + c];
        out[c] = acc;
    }
}

__global__ __launch_bounds__(256) void bias_grad_finalize_kernel(const float* __restrict__ partial, int nb, int C, int Cp, float* __restrict__ dbias) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double s, unused;
    partial_sums_16(partial, nb, Cp, c, s, unused);
    if ((threadIdx.x >> 4) == 0 && c < C) dbias[c] = (float)s;
}

static int cl_shape(long R, int C, ClShape* sh, int* nb) {
    if (R <= 0 || C <= 0) return ZSV_E_BAD_SHAPE;
    const int Cp = C <= 4 ? 4 : (C + 31) / 32 * 32;
    if (Cp < 32 || Cp / 8 > 256) return ZSV_E_UNSUPPORTED;
    if (R * Cp >= (1L << 40)) return ZSV_E_TOO_LARGE;
    sh->R = R; sh->Cp = Cp; sh->G = Cp / 8; sh->RL = 256 / sh->G;
    // about 1024 blocks (4 per CU), each at least 4 rows per row lane
    long rows = (R + 1023) / 1024;
    const long min_rows = 4L * sh->RL;
    if (rows < min_rows) rows = min_rows;
    rows = (rows + sh->RL - 1) / sh->RL * sh->RL;
    sh->rows_per_block = rows;
    *nb = (int)((R + rows - 1) / rows);
    return ZSV_OK;
}

static size_t cl_workspace_bytes(const ClShape& sh, int nb) { return ((size_t)nb * 2 * sh.Cp + 3 * (size_t)sh.Cp) * sizeof(float); }

}  // namespace zsv

using namespace zsv;

extern "C" {

size_t zsv_bn_cl_workspace_bytes(int64_t R, int32_t C) {
    ClShape sh;
    int nb = 0;
    if (cl_shape(R, C, &sh, &nb) != ZSV_OK) return 0;
    return cl_workspace_bytes(sh, nb);
}

static int bn_cl_fwd_train_impl(const void* z, const void* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int fuse_relu, void* y,
                                float* save_mean, float* save_invstd, float* save_coef, const float* conv_partials, int32_t conv_rows,
                                void* workspace, size_t workspace_bytes, void* stream);

int zsv_bn_cl_fwd_train(const void* z, const void* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, int fuse_relu, void* y,
                        float* save_mean, float* save_invstd, float* save_coef, void* workspace, size_t workspace_bytes, void* stream) {
    return bn_cl_fwd_train_impl(z, residual, R, C, gamma, beta, running_mean, running_var, momentum, eps, fuse_relu, y, save_mean,
                                save_invstd, save_coef, nullptr, 0, workspace, workspace_bytes, stream);
}

int zsv_bn_cl_fwd_train_stats(const void* z, const void* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int fuse_relu, void* y,
                              float* save_mean, float* save_invstd, float* save_coef, const float* conv_partials, int32_t conv_rows,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (conv_partials == nullptr || conv_rows <= 0) return ZSV_E_NULL;
    return bn_cl_fwd_train_impl(z, residual, R, C, gamma, beta, running_mean, running_var, momentum, eps, fuse_relu, y, save_mean,
                                save_invstd, save_coef, conv_partials, conv_rows, workspace, workspace_bytes, stream);
}

static int bn_cl_fwd_train_impl(const void* z, const void* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int fuse_relu, void* y,
                                float* save_mean, float* save_invstd, float* save_coef, const float* conv_partials, int32_t conv_rows,
                                void* workspace, size_t workspace_bytes, void* stream) {
    ClShape sh;
    int nb = 0;
    const int st = cl_shape(R, C, &sh, &nb);
    if (st != ZSV_OK) return st;
    if (!z || !y || !save_mean || !save_invstd || !workspace) return ZSV_E_NULL;
    if (workspace_bytes < cl_workspace_bytes(sh, nb)) return ZSV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* coef = save_coef ? save_coef : partial + (size_t)nb * 2 * sh.Cp;       // [2][Cp]: kept by the caller for the backward's mask
    const size_t lds = (size_t)sh.RL * 2 * sh.Cp * sizeof(float);
    if (conv_partials != nullptr) {      // the producing convolution's epilogue took the sums (zsv_conv3d_bf16_fwd_stats): same layout
        hipLaunchKernelGGL(bn_cl_fwd_finalize_kernel, dim3((sh.Cp + 15) / 16), dim3(256), 0, s, conv_partials, (int)conv_rows, C, sh.Cp,
                           (long)R, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, coef);
    } else {
        hipLaunchKernelGGL((bn_cl_reduce_kernel<0>), dim3(nb), dim3(256), lds, s, sh, (const u32x4v*)z, nullptr, nullptr, nullptr, partial);
        hipLaunchKernelGGL(bn_cl_fwd_finalize_kernel, dim3((sh.Cp + 15) / 16), dim3(256), 0, s, partial, nb, C, sh.Cp, (long)R, gamma, beta,
                           running_mean, running_var, momentum, eps, save_mean, save_invstd, coef);
    }
    const u32x4v* zz = (const u32x4v*)z;
    const u32x4v* rr = (const u32x4v*)residual;
    u32x4v* yy = (u32x4v*)y;
    if (residual) {
        if (fuse_relu) hipLaunchKernelGGL((bn_cl_apply_kernel<true, true>), dim3(nb), dim3(256), 0, s, sh, zz, rr, coef, yy);
        else hipLaunchKernelGGL((bn_cl_apply_kernel<true, false>), dim3(nb), dim3(256), 0, s, sh, zz, rr, coef, yy);
    } else {
        if (fuse_relu) hipLaunchKernelGGL((bn_cl_apply_kernel<false, true>), dim3(nb), dim3(256), 0, s, sh, zz, rr, coef, yy);
        else hipLaunchKernelGGL((bn_cl_apply_kernel<false, false>), dim3(nb), dim3(256), 0, s, sh, zz, rr, coef, yy);
    }
    return launch_status();
}

int zsv_bn_cl_bwd(const void* dy, const void* y, const void* z, int64_t R, int32_t C, const float* gamma, const float* save_mean,
                  const float* save_invstd, const float* fwd_coef, int relu_mask, void* dz, void* g_out, float* dgamma, float* dbeta,
                  void* workspace, size_t workspace_bytes, void* stream) {
    ClShape sh;
    int nb = 0;
    const int st = cl_shape(R, C, &sh, &nb);
    if (st != ZSV_OK) return st;
    if (!dy || !z || !dz || !save_mean || !save_invstd || !workspace) return ZSV_E_NULL;
    if (relu_mask && !y && !fwd_coef) return ZSV_E_NULL;
    if (workspace_bytes < cl_workspace_bytes(sh, nb)) return ZSV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* coef = partial + (size_t)nb * 2 * sh.Cp;
    const size_t lds = (size_t)sh.RL * 2 * sh.Cp * sizeof(float);
    const u32x4v *gg = (const u32x4v*)dy, *yy = (const u32x4v*)y, *zz = (const u32x4v*)z;
    const int mask = !relu_mask ? 0 : (y ? 1 : 2);          // 1: the saved output, 2: recomputed from z and the forward's coefficients
    if (mask == 1) hipLaunchKernelGGL((bn_cl_reduce_kernel<2>), dim3(nb), dim3(256), lds, s, sh, zz, gg, yy, fwd_coef, partial);
    else if (mask == 2) hipLaunchKernelGGL((bn_cl_reduce_kernel<3>), dim3(nb), dim3(256), lds, s, sh, zz, gg, yy, fwd_coef, partial);
    else hipLaunchKernelGGL((bn_cl_reduce_kernel<1>), dim3(nb), dim3(256), lds, s, sh, zz, gg, yy, fwd_coef, partial);
    hipLaunchKernelGGL(bn_cl_bwd_finalize_kernel, dim3((sh.Cp + 15) / 16), dim3(256), 0, s, partial, nb, C, sh.Cp, (long)R, gamma,
                       save_mean, save_invstd, dgamma, dbeta, coef);
    u32x4v *dd = (u32x4v*)dz, *go = (u32x4v*)g_out;
#define ZSV_BWD_APPLY(M, G) hipLaunchKernelGGL((bn_cl_bwd_apply_kernel<M, G>), dim3(nb), dim3(256), 0, s, sh, gg, yy, zz, coef, fwd_coef, dd, go)
    if (mask == 1) { if (g_out) ZSV_BWD_APPLY(1, true); else ZSV_BWD_APPLY(1, false); }
    else if (mask == 2) { if (g_out) ZSV_BWD_APPLY(2, true); else ZSV_BWD_APPLY(2, false); }
    else { if (g_out) ZSV_BWD_APPLY(0, true); else ZSV_BWD_APPLY(0, false); }
#undef ZSV_BWD_APPLY
    return launch_status();
}

int zsv_maxpool3d_bf16_bwd(const void* dy, const void* x, int32_t N, int32_t C, int32_t Ti, int32_t Hi, int32_t Wi, int32_t kT, int32_t kH,
                           int32_t kW, int32_t pT, int32_t pH, int32_t pW, int32_t To, int32_t Ho, int32_t Wo, void* dx, void* stream) {
    if (N <= 0 || C <= 4 || Ti <= 0 || Hi <= 0 || Wi <= 0 || kT <= 0 || kH <= 0 || kW <= 0 || pT < 0 || pH < 0 || pW < 0)
        return ZSV_E_BAD_SHAPE;
    if (2 * pT > kT || 2 * pH > kH || 2 * pW > kW) return ZSV_E_BAD_SHAPE;
    if (To != (Ti + 2 * pT - kT) / kT + 1 || Ho != (Hi + 2 * pH - kH) / kH + 1 || Wo != (Wi + 2 * pW - kW) / kW + 1 || To <= 0 ||
        Ho <= 0 || Wo <= 0)
        return ZSV_E_BAD_SHAPE;
    if (!dy || !x || !dx) return ZSV_E_NULL;
    const int G = (C + 31) / 32 * 32 / 8;
    if ((long)N * Ti * Hi * Wi * G >= (1L << 31)) return ZSV_E_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    // voxels past the last window (floor mode) receive no gradient: zero the tensor first only then
    if (To * kT - pT < Ti || Ho * kH - pH < Hi || Wo * kW - pW < Wi) {
        if (hipMemsetAsync(dx, 0, (size_t)N * Ti * Hi * Wi * G * 16, s) != hipSuccess) return ZSV_E_LAUNCH;
    }
    const long total = (long)N * To * Ho * Wo * G;
    hipLaunchKernelGGL(maxpool3d_cl_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const u32x4v*)dy, (const u32x4v*)x, Ti,
                       Hi, Wi, G, kT, kH, kW, pT, pH, pW, To, Ho, Wo, total, (u32x4v*)dx);
    return launch_status();
}

int zsv_relu_bias_bwd_cl(const void* dy, const void* y, int64_t R, int32_t C, void* g_out, float* dbias, void* workspace,
                         size_t workspace_bytes, void* stream) {
    ClShape sh;
    int nb = 0;
    const int st = cl_shape(R, C, &sh, &nb);
    if (st != ZSV_OK) return st;
    if (!dy || !y || !g_out || !workspace) return ZSV_E_NULL;
    if (workspace_bytes < cl_workspace_bytes(sh, nb)) return ZSV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const size_t lds = (size_t)sh.RL * 2 * sh.Cp * sizeof(float);
    hipLaunchKernelGGL(relu_bias_bwd_cl_kernel, dim3(nb), dim3(256), lds, s, sh, (const u32x4v*)dy, (const u32x4v*)y, (u32x4v*)g_out, partial);
    if (dbias) hipLaunchKernelGGL(bias_grad_finalize_kernel, dim3((sh.Cp + 15) / 16), dim3(256), 0, s, partial, nb, C, sh.Cp, dbias);
    return launch_status();
}

int zsv_cl_bf16_to_ncs_f32(const void* x, int32_t N, int32_t S, int32_t C, float* out, void* stream) {
    if (N <= 0 || S <= 0 || C <= 4) return ZSV_E_BAD_SHAPE;
    if (!x || !out) return ZSV_E_NULL;
    const int Cp = (C + 31) / 32 * 32;
    hipLaunchKernelGGL(cl_bf16_to_ncs_f32_kernel, dim3((S + 63) / 64, Cp / 32, N), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, S, C, Cp, out);
    return launch_status();
}

int zsv_ncs_f32_to_cl_bf16(const float* x, int32_t N, int32_t S, int32_t C, void* out, void* stream) {
    if (N <= 0 || S <= 0 || C <= 4) return ZSV_E_BAD_SHAPE;
    if (!x || !out) return ZSV_E_NULL;
    const int Cp = (C + 31) / 32 * 32;
    hipLaunchKernelGGL(ncs_f32_to_cl_bf16_kernel, dim3((S + 63) / 64, Cp / 32, N), dim3(256), 0, (hipStream_t)stream, x, S, C, Cp,
                       (unsigned short*)out);
    return launch_status();
}

int zsv_meanpool_bf16_bwd(const float* dpooled, int32_t N, int32_t S, int32_t C, void* dx, void* stream) {
    if (N <= 0 || S <= 0 || C <= 4) return ZSV_E_BAD_SHAPE;
    if (!dpooled || !dx) return ZSV_E_NULL;
    const int Cp = (C + 31) / 32 * 32;
    const long total = (long)N * S * Cp;
    hipLaunchKernelGGL(meanpool_cl_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dpooled, S, C,
                       Cp, total, (unsigned short*)dx);
    return launch_status();
}

}  // extern "C"
