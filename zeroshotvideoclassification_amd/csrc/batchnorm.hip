// batchnorm.hip -- BatchNorm3d forward/backward (+ fused residual add and ReLU), gfx950.
//
// Supplies aten::native_batch_norm / native_batch_norm_backward for every
// nn.BatchNorm3d of the reference (resnet.py:48,95,97,183,186,272), with the
// `out += residual; relu(out)` of BasicBlock.forward (resnet.py:110-111) and the ReLU of
// Conv2Plus1D / stems (resnet.py:49,95,184,187) fused into the same HBM pass.
//
// These are HBM-bound passes over (N, C, S) fp32 tensors (S = T*H*W contiguous):
//   stats   : one read of x            -> per (channel, slice) partial (sum, sumsq)
//   finalize: C threads                -> mean, invstd, scale/shift, running stats
//   apply   : read x (+res), write y   -> float4 along S when S % 4 == 0
//   bwd     : reduce (read dy, x, y) + apply (read dy, x, y; write dx (+dres))
// Per-thread partial sums are fp32 over <= ~128 elements, combined in fp64 with
// wave-level shuffles (64-wide) and a 4-entry LDS exchange; the slices of a channel are
// summed in fixed order, so results are bitwise reproducible run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "zsv_hip.h"
#include "zsv_common.h"
#include "knobs.h"

namespace zsv {

// Streaming accesses with the non-temporal hint (`nt`, wave-uniform): the big activation tensors are read once per pass, and their
// lines only push out what the kernels running next to these passes still need (the side queue's weight-gradient kernels and
// their weight panels).  Threshold sweep inside the training step (one device, two rounds each): off 39.06 / 39.13 ms, 200 MB
// 38.87 / 38.89, 100 MB 38.88 / 38.87 (38.78 / 38.76 on another device), 30 MB 38.79 / 38.78, every tensor 38.83 / 38.84.
typedef float bn_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p, bool nt) {
    if (nt) {
        const bn_v4f v = __builtin_nontemporal_load(reinterpret_cast<const bn_v4f*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, const float4& v, bool nt) {
    if (nt) {
        const bn_v4f w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<bn_v4f*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}
// bytes of ONE tensor of the pass above which its accesses are streamed (ZSV_BN_NT_MB: default 64; 0 = never)
static inline bool bn_stream(int N, int C, int S) {
    const char* e = ZSV_KNOB(BN_NT_MB);
    const double mb = e ? atof(e) : 64.0;
    return mb > 0.0 && (double)N * C * S * 4.0 > mb * 1e6;
}


// slices per channel so that every channel has enough workgroups in flight and a thread
// accumulates a bounded number of fp32 terms
static inline int bn_slices(int N, int C, int S) {
    const long per_channel = (long)N * S;
    long want = (2048 + C - 1) / C;                 // ~8 workgroups per CU overall
    long by_len = (per_channel + 256L * 128 - 1) / (256L * 128);   // <= 128 terms per thread
    long s = want > by_len ? want : by_len;
    long max_s = (per_channel + 1023) / 1024;       // at least 1024 elements per slice
    if (max_s < 1) max_s = 1;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 4096) s = 4096;
    return (int)s;
}

// workspace layout: double part[2][C][slices] | float scale[C] | float shift[C] | float c1[C] | float c2[C]
struct BnWs {
    double* part;
    float* scale;
    float* shift;
    float* c1;
    float* c2;
};
static inline size_t bn_ws_bytes(int N, int C, int S) {
    const int sl = bn_slices(N, C, S);
    return (size_t)2 * C * sl * sizeof(double) + (size_t)4 * C * sizeof(float);
}
static inline BnWs bn_ws(void* ws, int C, int slices) {
    BnWs w;
    w.part = (double*)ws;
    w.scale = (float*)(w.part + (size_t)2 * C * slices);
    w.shift = w.scale + C;
    w.c1 = w.shift + C;
    w.c2 = w.c1 + C;
    return w;
}

// The (n, s) domain of one channel is N*S elements; slice `sl` owns the contiguous range
// [sl*len, (sl+1)*len) of it, len a multiple of 4 when S % 4 == 0.
__device__ __forceinline__ void slice_range(int total, int slices, int sl, bool vec, int& b, int& e) {
    int len = (total + slices - 1) / slices;
    if (vec) len = (len + 3) & ~3;
    const long bl = (long)sl * len;
    b = bl > total ? total : (int)bl;
    e = (total - b < len) ? total : b + len;
}

// ---- forward statistics ------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int N, int C, int S, int slices,
                                                       double* __restrict__ part) {
    __shared__ double red[4];
    const int c = blockIdx.x, sl = blockIdx.y;
    const bool vec = (S % 4) == 0;
    int b, e;
    slice_range(N * S, slices, sl, vec, b, e);
    // Shifted sums: a sample of the channel is the pivot K, so sum (x-K)^2 - (sum (x-K))^2 / n does not cancel
    // when |mean| >> std (biased convolutions, post-ReLU features); x - K is exact or 1-ulp in fp32.
    const float K = x[(size_t)c * S];
    float s1 = 0.f, s2 = 0.f;
    if (vec) {
        for (int i = b + 4 * (int)threadIdx.x; i < e; i += 4 * 256) {
            const int n = i / S, s = i - n * S;
            float4 v = *reinterpret_cast<const float4*>(x + ((size_t)n * C + c) * S + s);
            v.x -= K; v.y -= K; v.z -= K; v.w -= K;
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (int i = b + (int)threadIdx.x; i < e; i += 256) {
            const int n = i / S, s = i - n * S;
            const float v = x[((size_t)n * C + c) * S + s] - K;
            s1 += v;
            s2 += v * v;
        }
    }
    const double t1 = block_sum_256<double>((double)s1, red);
    const double t2 = block_sum_256<double>((double)s2, red);
    if (threadIdx.x == 0) {
        part[(size_t)c * slices + sl] = t1;
        part[(size_t)(C + c) * slices + sl] = t2;
    }
}

// `pivot_src` (may be null = pivot 0): the tensor bn_stats_kernel shifted by x[c * S]; the partial sums are then
// sums of (x - K) and (x - K)^2.
__global__ void bn_finalize_train_kernel(const double* __restrict__ part, int C, int slices, double count,
                                         const float* __restrict__ pivot_src, int S,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         float momentum, float eps, float* __restrict__ scale,
                                         float* __restrict__ shift) {
    const int c = blockIdx.x;                      // one 64-lane wave per channel (as bn_bwd_finalize_kernel)
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < slices; k += 64) {
        s1 += part[(size_t)c * slices + k];
        s2 += part[(size_t)(C + c) * slices + k];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (threadIdx.x != 0) return;
    const double K = pivot_src ? (double)pivot_src[(size_t)c * S] : 0.0;
    const double dmean = s1 / count;                 // mean - K
    const double mean = K + dmean;
    double var = s2 / count - dmean * dmean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = g * (float)invstd;
    scale[c] = sc;
    shift[c] = __fmaf_rn(-(float)mean, sc, bt);      // (the backward recomputes exactly this)
}

// Statistics from the producing convolution's epilogue partials ([C][tiles] sums and [C][tiles] sums of squares, fp32) and the
// finalisation in ONE launch (a block per channel): a double-precision block sum in a fixed order, then exactly
// bn_finalize_train_kernel's arithmetic with pivot 0.
__global__ __launch_bounds__(256) void bn_finalize_partials_kernel(const float* __restrict__ psum, const float* __restrict__ psq,
                                                                   int C, int tiles, double count, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ save_mean,
                                                                   float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                                   float* __restrict__ running_var, float momentum, float eps,
                                                                   float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    double a1 = 0.0, a2 = 0.0;
    for (int t = threadIdx.x; t < tiles; t += 256) {
        a1 += (double)psum[(size_t)c * tiles + t];
        a2 += (double)psq[(size_t)c * tiles + t];
    }
    const double s1 = block_sum_256<double>(a1, red);
    const double s2 = block_sum_256<double>(a2, red);
    if (threadIdx.x != 0) return;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = g * (float)invstd;
    scale[c] = sc;
    shift[c] = __fmaf_rn(-(float)mean, sc, bt);
}

__global__ void bn_eval_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                     float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = bt - rm[c] * sc;
}

// ---- y = relu?(x*scale[c] + shift[c] + res?) ------------------------------------------
// grid.x = row (n*C + c), grid.y = chunk of the row; 256 threads x 16 elements per chunk.
constexpr int ROW_CHUNK = 4096;

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       float* __restrict__ y, int C, int S,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, bool nt) {
    const int row = blockIdx.x;
    const int c = row % C;
    const float sc = scale[c], sh = shift[c];
    const size_t base = (size_t)row * S;
    const int s0 = blockIdx.y * ROW_CHUNK;
    const int s1 = min(S, s0 + ROW_CHUNK);
    if ((S % 4) == 0 && s1 - s0 == ROW_CHUNK) {
        // full chunk: all 4 (8 with a residual) 16-byte loads of a thread are issued before the first use
        float4 v[4], r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ld4(x + base + s0 + 4 * threadIdx.x + 1024 * i, nt);
        if (RES) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = ld4(res + base + s0 + 4 * threadIdx.x + 1024 * i, nt);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i].x = __fmaf_rn(v[i].x, sc, sh); v[i].y = __fmaf_rn(v[i].y, sc, sh);
            v[i].z = __fmaf_rn(v[i].z, sc, sh); v[i].w = __fmaf_rn(v[i].w, sc, sh);
            if (RES) { v[i].x += r[i].x; v[i].y += r[i].y; v[i].z += r[i].z; v[i].w += r[i].w; }
            if (RELU) { v[i].x = fmaxf(v[i].x, 0.f); v[i].y = fmaxf(v[i].y, 0.f); v[i].z = fmaxf(v[i].z, 0.f); v[i].w = fmaxf(v[i].w, 0.f); }
            st4(y + base + s0 + 4 * threadIdx.x + 1024 * i, v[i], nt);
        }
    } else if ((S % 4) == 0) {
        for (int s = s0 + 4 * threadIdx.x; s < s1; s += 1024) {
            float4 v = *reinterpret_cast<const float4*>(x + base + s);
            v.x = __fmaf_rn(v.x, sc, sh); v.y = __fmaf_rn(v.y, sc, sh); v.z = __fmaf_rn(v.z, sc, sh); v.w = __fmaf_rn(v.w, sc, sh);
            if (RES) {
                const float4 r = *reinterpret_cast<const float4*>(res + base + s);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(y + base + s) = v;
        }
    } else {
        for (int s = s0 + threadIdx.x; s < s1; s += 256) {
            float v = __fmaf_rn(x[base + s], sc, sh);
            if (RES) v += res[base + s];
            if (RELU) v = fmaxf(v, 0.f);
            y[base + s] = v;
        }
    }
}

// ---- backward reduce: sum g, sum g*xhat with g = dy * mask ------------------------------
// RELU: 0 no mask; 1 mask = (saved output y > 0); 2 mask = (x*scale + shift > 0), i.e. the
// pre-activation recomputed with the forward's exact fma -- saves reading y when the forward
// had no residual input.
template <int RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ y, int N, int C, int S,
                                                            int slices, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            double* __restrict__ part, float* __restrict__ gout, bool nt) {
    // gout != nullptr (RELU 1 with a residual branch): the masked gradient g = dy * (y > 0) is written here -- it IS the gradient of
    // the residual branch -- and the apply pass then reads g alone instead of dy and y again (8 -> 7 passes over the tensor)
    __shared__ double red[4];
    const int c = blockIdx.x, sl = blockIdx.y;
    const bool vec = (S % 4) == 0;
    int b, e;
    slice_range(N * S, slices, sl, vec, b, e);
    const float mu = mean[c], is = invstd[c];
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    const float sh = __fmaf_rn(-mu, sc, beta ? beta[c] : 0.f);
    float s1 = 0.f, s2 = 0.f;
    if (vec) {
        for (int i = b + 4 * (int)threadIdx.x; i < e; i += 4 * 256) {
            const int n = i / S, s = i - n * S;
            const size_t off = ((size_t)n * C + c) * S + s;
            float4 g = ld4(dy + off, nt);
            const float4 xv = ld4(x + off, nt);
            if (RELU == 1) {
                const float4 yv = ld4(y + off, nt);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            } else if (RELU == 2) {
                g.x = __fmaf_rn(xv.x, sc, sh) > 0.f ? g.x : 0.f; g.y = __fmaf_rn(xv.y, sc, sh) > 0.f ? g.y : 0.f;
                g.z = __fmaf_rn(xv.z, sc, sh) > 0.f ? g.z : 0.f; g.w = __fmaf_rn(xv.w, sc, sh) > 0.f ? g.w : 0.f;
            }
            if (RELU == 1 && gout) *reinterpret_cast<float4*>(gout + off) = g;
            s1 += (g.x + g.y) + (g.z + g.w);
            s2 += (g.x * ((xv.x - mu) * is) + g.y * ((xv.y - mu) * is)) +
                  (g.z * ((xv.z - mu) * is) + g.w * ((xv.w - mu) * is));
        }
    } else {
        for (int i = b + (int)threadIdx.x; i < e; i += 256) {
            const int n = i / S, s = i - n * S;
            const size_t off = ((size_t)n * C + c) * S + s;
            float g = dy[off];
            if (RELU == 1) g = y[off] > 0.f ? g : 0.f;
            else if (RELU == 2) g = __fmaf_rn(x[off], sc, sh) > 0.f ? g : 0.f;
            if (RELU == 1 && gout) gout[off] = g;
            s1 += g;
            s2 += g * ((x[off] - mu) * is);
        }
    }
    const double t1 = block_sum_256<double>((double)s1, red);
    const double t2 = block_sum_256<double>((double)s2, red);
    if (threadIdx.x == 0) {
        part[(size_t)c * slices + sl] = t1;
        part[(size_t)(C + c) * slices + sl] = t2;
    }
}

// dgamma = sum g*xhat, dbeta = sum g;  dx = a*g + b*x + k  with
//   a = gamma*invstd, b = -a*invstd*dgamma/M, k = -a*dbeta/M - b*mean
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int C, int slices, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ ca, float* __restrict__ cb,
                                       float* __restrict__ ck, const float* __restrict__ beta,
                                       float* __restrict__ csh) {
    // one 64-lane wave per channel: lane l adds slices l, l + 64, ... in order, then the lanes are combined by xor-shuffles -- a
    // fixed order (bitwise reproducible); a single thread walking up to ~35 slices with dependent adds took 16 us per launch,
    // 37 launches per step on the backward's critical path
    const int c = blockIdx.x;
    double sg = 0.0, sgx = 0.0;
    for (int k = threadIdx.x; k < slices; k += 64) {
        sg += part[(size_t)c * slices + k];
        sgx += part[(size_t)(C + c) * slices + k];
    }
    sg = wave_sum(sg);
    sgx = wave_sum(sgx);
    if (threadIdx.x != 0) return;
    if (dgamma) dgamma[c] = (float)sgx;
    if (dbeta) dbeta[c] = (float)sg;
    const double g = gamma ? (double)gamma[c] : 1.0;
    const double is = (double)invstd[c], mu = (double)mean[c];
    const double a = g * is;
    const double b = -a * is * sgx / count;
    const double k = -a * sg / count - b * mu;
    ca[c] = (float)a;
    cb[c] = (float)b;
    ck[c] = (float)k;
    // forward's shift, recomputed with the forward's exact arithmetic (its scale is gamma*invstd in fp32)
    const float sc = (gamma ? gamma[c] : 1.f) * invstd[c];
    csh[c] = __fmaf_rn(-mean[c], sc, beta ? beta[c] : 0.f);
}

template <int RELU, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, float* __restrict__ dx,
                                                           float* __restrict__ dres, int C, int S,
                                                           const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const float* __restrict__ ck, const float* __restrict__ gamma,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ csh, bool nt) {
    const int row = blockIdx.x;
    const int c = row % C;
    const float a = ca[c], b = cb[c], k = ck[c];
    const float sc = RELU == 2 ? (gamma ? gamma[c] : 1.f) * invstd[c] : 0.f;
    const float sh = RELU == 2 ? csh[c] : 0.f;
    const size_t base = (size_t)row * S;
    const int s0 = blockIdx.y * ROW_CHUNK;
    const int s1 = min(S, s0 + ROW_CHUNK);
    if ((S % 4) == 0 && s1 - s0 == ROW_CHUNK) {
        // full chunk: the 8 (12 with a saved-output mask) 16-byte loads of a thread are issued up front
        float4 gv[4], xq[4], yq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gv[i] = ld4(dy + base + s0 + 4 * threadIdx.x + 1024 * i, nt);
            xq[i] = ld4(x + base + s0 + 4 * threadIdx.x + 1024 * i, nt);
            if (RELU == 1) yq[i] = ld4(y + base + s0 + 4 * threadIdx.x + 1024 * i, nt);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 g = gv[i];
            const float4 xv = xq[i];
            const int s = s0 + 4 * threadIdx.x + 1024 * i;
            if (RELU == 1) {
                const float4 yv = yq[i];
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            } else if (RELU == 2) {
                g.x = __fmaf_rn(xv.x, sc, sh) > 0.f ? g.x : 0.f; g.y = __fmaf_rn(xv.y, sc, sh) > 0.f ? g.y : 0.f;
                g.z = __fmaf_rn(xv.z, sc, sh) > 0.f ? g.z : 0.f; g.w = __fmaf_rn(xv.w, sc, sh) > 0.f ? g.w : 0.f;
            }
            if (DRES) st4(dres + base + s, g, nt);
            float4 o;
            o.x = a * g.x + b * xv.x + k; o.y = a * g.y + b * xv.y + k;
            o.z = a * g.z + b * xv.z + k; o.w = a * g.w + b * xv.w + k;
            st4(dx + base + s, o, nt);
        }
    } else if ((S % 4) == 0) {
        for (int s = s0 + 4 * threadIdx.x; s < s1; s += 1024) {
            float4 g = *reinterpret_cast<const float4*>(dy + base + s);
            const float4 xv = *reinterpret_cast<const float4*>(x + base + s);
            if (RELU == 1) {
                const float4 yv = *reinterpret_cast<const float4*>(y + base + s);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            } else if (RELU == 2) {
                g.x = __fmaf_rn(xv.x, sc, sh) > 0.f ? g.x : 0.f; g.y = __fmaf_rn(xv.y, sc, sh) > 0.f ? g.y : 0.f;
                g.z = __fmaf_rn(xv.z, sc, sh) > 0.f ? g.z : 0.f; g.w = __fmaf_rn(xv.w, sc, sh) > 0.f ? g.w : 0.f;
            }
            if (DRES) *reinterpret_cast<float4*>(dres + base + s) = g;
            float4 o;
            o.x = a * g.x + b * xv.x + k; o.y = a * g.y + b * xv.y + k;
            o.z = a * g.z + b * xv.z + k; o.w = a * g.w + b * xv.w + k;
            *reinterpret_cast<float4*>(dx + base + s) = o;
        }
    } else {
        for (int s = s0 + threadIdx.x; s < s1; s += 256) {
            float g = dy[base + s];
            const float xs = x[base + s];
            if (RELU == 1) g = y[base + s] > 0.f ? g : 0.f;
            else if (RELU == 2) g = __fmaf_rn(xs, sc, sh) > 0.f ? g : 0.f;
            if (DRES) dres[base + s] = g;
            dx[base + s] = a * g + b * xs + k;
        }
    }
}

static int check_ncs(int N, int C, int S) {
    if (N <= 0 || C <= 0 || S <= 0) return ZSV_E_BAD_SHAPE;
    if ((double)N * C * S >= 2147483647.0) return ZSV_E_TOO_LARGE;
    return ZSV_OK;
}

static int launch_apply(const float* x, const float* res, float* y, int N, int C, int S, const float* scale,
                        const float* shift, int relu, hipStream_t stream) {
    const dim3 grid((unsigned)(N * C), (unsigned)((S + ROW_CHUNK - 1) / ROW_CHUNK));
    const bool nt = bn_stream(N, C, S);
    if (res && relu) hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, dim3(256), 0, stream, x, res, y, C, S, scale, shift, nt);
    else if (res) hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, dim3(256), 0, stream, x, res, y, C, S, scale, shift, nt);
    else if (relu) hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, dim3(256), 0, stream, x, res, y, C, S, scale, shift, nt);
    else hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, dim3(256), 0, stream, x, res, y, C, S, scale, shift, nt);
    return launch_status();
}

}  // namespace zsv

using namespace zsv;

extern "C" size_t zsv_bn_workspace_bytes(int32_t N, int32_t C, int32_t S) {
    if (check_ncs(N, C, S) != ZSV_OK) return 0;
    return bn_ws_bytes(N, C, S);
}

extern "C" int zsv_bn_fwd_train(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma,
                                const float* beta, const float* residual, int fuse_relu, float* y,
                                float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                float momentum, float eps, void* workspace, size_t workspace_bytes, void* stream_) {
    return zsv_bn_fwd_train_stats(x, N, C, S, gamma, beta, residual, fuse_relu, y, save_mean, save_invstd, running_mean,
                                  running_var, momentum, eps, nullptr, 0, workspace, workspace_bytes, stream_);
}

extern "C" int zsv_bn_fwd_train_stats(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma,
                                      const float* beta, const float* residual, int fuse_relu, float* y,
                                      float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                      float momentum, float eps, const float* conv_partials, int32_t stat_tiles,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    int st = check_ncs(N, C, S);
    if (st) return st;
    if (!x || !y || !save_mean || !save_invstd || !workspace) return ZSV_E_NULL;
    if (workspace_bytes < bn_ws_bytes(N, C, S)) return ZSV_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    int slices = bn_slices(N, C, S);
    BnWs w = bn_ws(workspace, C, slices);
    const float* pivot_src = x;
    if (conv_partials && stat_tiles > 0) {
        // statistics were accumulated by the producing convolution's epilogue (unshifted sums: bias-free convolution outputs,
        // |mean| ~ std): no pass over x, reduce + finalise in one launch
        hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3(C), dim3(256), 0, stream, conv_partials,
                           conv_partials + (size_t)C * stat_tiles, C, stat_tiles, (double)N * S, gamma, beta, save_mean, save_invstd,
                           running_mean, running_var, momentum, eps, w.scale, w.shift);
    } else {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, slices), dim3(256), 0, stream, x, N, C, S, slices, w.part);
        if ((st = launch_status())) return st;
        hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C), dim3(64), 0, stream, (const double*)w.part, C,
                           slices, (double)N * S, pivot_src, S, gamma, beta, save_mean, save_invstd, running_mean, running_var,
                           momentum, eps, w.scale, w.shift);
    }
    if ((st = launch_status())) return st;
    return launch_apply(x, residual, y, N, C, S, w.scale, w.shift, fuse_relu, stream);
}

// Training-mode BatchNorm WITHOUT the normalise pass: statistics (own pass or the producing convolution's epilogue partials),
// save_mean / save_invstd, running statistics, and the per-channel affine y = x * scale + shift written to `coef`
// ([2][coef_pitch]: scale row, shift row; entries C .. coef_pitch-1 are set to 0).  The consumer applies it (+ ReLU) while
// it reads x: zsv_conv3d_fwd_pre / zsv_conv3d_wgrad_pre.  The backward is zsv_bn_bwd with relu_mode 2 (mask recomputed from x).
extern "C" int zsv_bn_fwd_train_coeffs(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma, const float* beta,
                                       float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                       float momentum, float eps, const float* conv_partials, int32_t stat_tiles, float* coef,
                                       int32_t coef_pitch, void* workspace, size_t workspace_bytes, void* stream_) {
    int st = check_ncs(N, C, S);
    if (st) return st;
    if (!x || !save_mean || !save_invstd || !workspace || !coef) return ZSV_E_NULL;
    if (coef_pitch < C) return ZSV_E_BAD_SHAPE;
    if (workspace_bytes < bn_ws_bytes(N, C, S)) return ZSV_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    int slices = bn_slices(N, C, S);
    BnWs w = bn_ws(workspace, C, slices);
    const float* pivot_src = x;
    if (coef_pitch > C && hipMemsetAsync(coef, 0, (size_t)2 * coef_pitch * sizeof(float), stream) != hipSuccess) return ZSV_E_LAUNCH;
    if (conv_partials && stat_tiles > 0) {
        hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3(C), dim3(256), 0, stream, conv_partials,
                           conv_partials + (size_t)C * stat_tiles, C, stat_tiles, (double)N * S, gamma, beta, save_mean, save_invstd,
                           running_mean, running_var, momentum, eps, coef, coef + coef_pitch);
        return launch_status();
    }
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, slices), dim3(256), 0, stream, x, N, C, S, slices, w.part);
    if ((st = launch_status())) return st;
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C), dim3(64), 0, stream, (const double*)w.part, C,
                       slices, (double)N * S, pivot_src, S, gamma, beta, save_mean, save_invstd, running_mean, running_var,
                       momentum, eps, coef, coef + coef_pitch);
    return launch_status();
}

extern "C" int zsv_bn_fwd_eval(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma, const float* beta,
                               const float* running_mean, const float* running_var, const float* residual,
                               int fuse_relu, float eps, float* y, void* workspace, size_t workspace_bytes,
                               void* stream_) {
    int st = check_ncs(N, C, S);
    if (st) return st;
    if (!x || !y || !running_mean || !running_var || !workspace) return ZSV_E_NULL;
    if (workspace_bytes < bn_ws_bytes(N, C, S)) return ZSV_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    BnWs w = bn_ws(workspace, C, bn_slices(N, C, S));
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 127) / 128), dim3(128), 0, stream, C, gamma, beta, running_mean,
                       running_var, eps, w.scale, w.shift);
    if ((st = launch_status())) return st;
    return launch_apply(x, residual, y, N, C, S, w.scale, w.shift, fuse_relu, stream);
}

extern "C" int zsv_bn_bwd(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
                          const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                          int fuse_relu, float* dx, float* d_residual, float* dgamma, float* dbeta, void* workspace,
                          size_t workspace_bytes, void* stream_) {
    int st = check_ncs(N, C, S);
    if (st) return st;
    if (!dy || !x || !dx || !save_mean || !save_invstd || !workspace) return ZSV_E_NULL;
    if (fuse_relu < 0 || fuse_relu > 2) return ZSV_E_BAD_SHAPE;
    if (fuse_relu == 1 && !y) return ZSV_E_NULL;
    if (workspace_bytes < bn_ws_bytes(N, C, S)) return ZSV_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int slices = bn_slices(N, C, S);
    BnWs w = bn_ws(workspace, C, slices);
    const dim3 rgrid(C, slices);
    // fuse_relu 1 with a residual gradient: the reduction writes the masked gradient (= d_residual), the apply pass reads it (mode 0)
    const bool g_from_reduce = fuse_relu == 1 && d_residual != nullptr && ZSV_KNOB(BN_NO_MASKED_G) == nullptr;
    const bool nt = bn_stream(N, C, S);
    if (fuse_relu == 2)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<2>), rgrid, dim3(256), 0, stream, dy, x, y, N, C, S, slices, save_mean, save_invstd, gamma, beta, w.part, (float*)nullptr, nt);
    else if (fuse_relu == 1)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<1>), rgrid, dim3(256), 0, stream, dy, x, y, N, C, S, slices, save_mean, save_invstd, gamma, beta, w.part,
                           g_from_reduce ? d_residual : (float*)nullptr, nt);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<0>), rgrid, dim3(256), 0, stream, dy, x, y, N, C, S, slices, save_mean, save_invstd, gamma, beta, w.part, (float*)nullptr, nt);
    if ((st = launch_status())) return st;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, stream, (const double*)w.part, C, slices,
                       (double)N * S, gamma, save_mean, save_invstd, dgamma, dbeta, w.scale, w.shift, w.c1, beta, w.c2);
    if ((st = launch_status())) return st;
    const dim3 grid((unsigned)(N * C), (unsigned)((S + ROW_CHUNK - 1) / ROW_CHUNK));
#define ZSV_BWD_APPLY(R, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<R, D>), grid, dim3(256), 0, stream, dy, x, y, dx, d_residual, C, S, w.scale, w.shift, w.c1, gamma, save_invstd, w.c2, nt)
    if (g_from_reduce) {
        hipLaunchKernelGGL((bn_bwd_apply_kernel<0, false>), grid, dim3(256), 0, stream, (const float*)d_residual, x, y, dx, (float*)nullptr, C, S, w.scale, w.shift, w.c1,
                           gamma, save_invstd, w.c2, nt);
        return launch_status();
    }
    if (fuse_relu == 2) { if (d_residual) ZSV_BWD_APPLY(2, true); else ZSV_BWD_APPLY(2, false); }
    else if (fuse_relu == 1) { if (d_residual) ZSV_BWD_APPLY(1, true); else ZSV_BWD_APPLY(1, false); }
    else { if (d_residual) ZSV_BWD_APPLY(0, true); else ZSV_BWD_APPLY(0, false); }
#undef ZSV_BWD_APPLY
    return launch_status();
}
