// elementwise_pool.hip -- ReLU, residual add, pooling, channel sums and Adam for gfx950.
//
// HBM-bound passes that the reference reaches through torch.nn:
//   relu        nn.ReLU / F.relu          resnet.py:49,95,98  network.py:147-166,614
//   add_relu    out += residual; relu     resnet.py:110-111
//   meanpool    torch.mean(f,(2,3,4))     network.py:595 (AdaptiveAvgPool3d resnet.py:251)
//   maxpool3d   nn.MaxPool3d              network.py:103-118 (C3D)
//   channel_sum bias gradient             network.py:102-117 convs, nn.Linear biases
//   adam        torch.optim.Adam.step     main.py:131,200
// Everything is float4-wide along the contiguous axis where alignment allows, grid-strided
// with at most 2048 workgroups (8 per CU), reductions use 64-lane wave shuffles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zsv_hip.h"
#include "zsv_common.h"

namespace zsv {

static inline unsigned ew_blocks(long work_items) {
    long b = (work_items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int OP>   // 0 relu fwd (a), 1 relu bwd (a = dy, b = y), 2 add_relu (a + b)
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, long n, int vec) {
    const long stride = (long)gridDim.x * 256;
    if (vec) {
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 x = reinterpret_cast<const float4*>(a)[i];
            float4 r;
            if (OP == 0) {
                r = make_float4(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f), fmaxf(x.w, 0.f));
            } else {
                const float4 y = reinterpret_cast<const float4*>(b)[i];
                if (OP == 1)
                    r = make_float4(y.x > 0.f ? x.x : 0.f, y.y > 0.f ? x.y : 0.f, y.z > 0.f ? x.z : 0.f, y.w > 0.f ? x.w : 0.f);
                else
                    r = make_float4(fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f));
            }
            reinterpret_cast<float4*>(out)[i] = r;
        }
        for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float x = a[i];
            out[i] = OP == 0 ? fmaxf(x, 0.f) : (OP == 1 ? (b[i] > 0.f ? x : 0.f) : fmaxf(x + b[i], 0.f));
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float x = a[i];
            out[i] = OP == 0 ? fmaxf(x, 0.f) : (OP == 1 ? (b[i] > 0.f ? x : 0.f) : fmaxf(x + b[i], 0.f));
        }
    }
}

// one wave per (n, c) row: mean over S
__global__ __launch_bounds__(256) void meanpool_fwd_kernel(const float* __restrict__ x, int rows, int S,
                                                           float* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + (size_t)row * S;
    float s = 0.f;
    for (int i = lane; i < S; i += 64) s += p[i];
    s = wave_sum(s);
    if (lane == 0) y[row] = s / (float)S;
}

__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ dy, long total, int S,
                                                           float* __restrict__ dx) {
    const float inv = 1.f / (float)S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        dx[i] = dy[i / S] * inv;
}

struct PoolGeom {
    int Ti, Hi, Wi, To, Ho, Wo, kT, kH, kW, pT, pH, pW;
};

// one thread per output voxel; scan order (t, h, w), first maximum wins, NaN propagates
// (same rule as aten's max_pool3d_with_indices)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, long total_out, PoolGeom g,
                                                          float* __restrict__ y, int* __restrict__ arg) {
    const int oS = g.To * g.Ho * g.Wo;
    const int iS = g.Ti * g.Hi * g.Wi;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total_out; o += (long)gridDim.x * 256) {
        const long row = o / oS;
        int r = (int)(o - row * oS);
        const int ot = r / (g.Ho * g.Wo);
        r -= ot * g.Ho * g.Wo;
        const int oh = r / g.Wo, ow = r - oh * g.Wo;
        const float* p = x + row * iS;
        float best = -INFINITY;
        int besti = -1;
        for (int a = 0; a < g.kT; ++a) {
            const int t = ot * g.kT - g.pT + a;
            if ((unsigned)t >= (unsigned)g.Ti) continue;
            for (int b = 0; b < g.kH; ++b) {
                const int h = oh * g.kH - g.pH + b;
                if ((unsigned)h >= (unsigned)g.Hi) continue;
                for (int c = 0; c < g.kW; ++c) {
                    const int w = ow * g.kW - g.pW + c;
                    if ((unsigned)w >= (unsigned)g.Wi) continue;
                    const int idx = (t * g.Hi + h) * g.Wi + w;
                    const float v = p[idx];
                    if (besti < 0) besti = idx;           // aten starts at the window's first voxel
                    if (v > best || v != v) { best = v; besti = idx; }
                }
            }
        }
        y[o] = best;
        arg[o] = besti;
    }
}

// windows do not overlap (kernel == stride): every input voxel belongs to at most one
// window, so dx is a gather -- no atomics
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg,
                                                          long total_in, PoolGeom g, float* __restrict__ dx) {
    const int oS = g.To * g.Ho * g.Wo;
    const int iS = g.Ti * g.Hi * g.Wi;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_in; i += (long)gridDim.x * 256) {
        const long row = i / iS;
        const int idx = (int)(i - row * iS);
        int r = idx;
        const int t = r / (g.Hi * g.Wi);
        r -= t * g.Hi * g.Wi;
        const int h = r / g.Wi, w = r - h * g.Wi;
        const int ot = (t + g.pT) / g.kT, oh = (h + g.pH) / g.kH, ow = (w + g.pW) / g.kW;
        float v = 0.f;
        if (ot < g.To && oh < g.Ho && ow < g.Wo) {
            const long o = row * oS + (long)(ot * g.Ho + oh) * g.Wo + ow;
            if (arg[o] == idx) v = dy[o];
        }
        dx[i] = v;
    }
}

// The same gather with four consecutive voxels of a W row per thread (Wi % 4 == 0): one (t, h) decode and one 16-byte store
// per quad, the (n, c) row on grid.y -- the scalar form above spends its time in per-voxel integer divisions
// (1.16 ms per launch on C3D's pools against 0.2-0.35 ms of HBM time).
__global__ __launch_bounds__(256) void maxpool_bwd_quad_kernel(const float* __restrict__ dy, const int* __restrict__ arg,
                                                               int rows, PoolGeom g, float* __restrict__ dx) {
    const unsigned oS = g.To * g.Ho * g.Wo, iS = g.Ti * g.Hi * g.Wi, HW = g.Hi * g.Wi, quads = iS >> 2;
    for (unsigned row = blockIdx.y; row < (unsigned)rows; row += gridDim.y) {
        const float* dyr = dy + (size_t)row * oS;
        const int* argr = arg + (size_t)row * oS;
        float* dxr = dx + (size_t)row * iS;
        for (unsigned q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256) {
            const unsigned idx = 4 * q;
            const unsigned t = idx / HW, r = idx - t * HW;
            const unsigned h = r / (unsigned)g.Wi, w = r - h * (unsigned)g.Wi;
            const unsigned ot = (t + g.pT) / (unsigned)g.kT, oh = (h + g.pH) / (unsigned)g.kH;
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ot < (unsigned)g.To && oh < (unsigned)g.Ho) {
                const unsigned obase = (ot * g.Ho + oh) * g.Wo;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned ow = (w + e + g.pW) / (unsigned)g.kW;
                    v[e] = 0.f;
                    if (ow < (unsigned)g.Wo) {
                        const unsigned o = obase + ow;
                        if (argr[o] == (int)(idx + e)) v[e] = dyr[o];
                    }
                }
                out = make_float4(v[0], v[1], v[2], v[3]);
            }
            *reinterpret_cast<float4*>(dxr + idx) = out;
        }
    }
}

// channel sum over (n, s): partial per (c, slice) in double, then a fixed-order combine
static inline int cs_slices(int N, int C, int S) {
    const long per = (long)N * S;
    long s = (1024 + C - 1) / C;
    long mx = (per + 1023) / 1024;
    if (mx < 1) mx = 1;
    if (s > mx) s = mx;
    if (s < 1) s = 1;
    return (int)s;
}

__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dy, int N, int C, int S,
                                                          int slices, double* __restrict__ part) {
    __shared__ double red[4];
    const int c = blockIdx.x, sl = blockIdx.y;
    const int total = N * S;
    const int len = (total + slices - 1) / slices;
    const int b = sl * len;
    const int e = min(total, b + len);
    float s1 = 0.f;
    for (int i = b + (int)threadIdx.x; i < e; i += 256) {
        const int n = i / S, s = i - n * S;
        s1 += dy[((size_t)n * C + c) * S + s];
    }
    const double t = block_sum_256<double>((double)s1, red);
    if (threadIdx.x == 0) part[(size_t)c * slices + sl] = t;
}

// relu'(y) * dy and its per-channel sum in one pass (C3D's `relu(conv(x) + b)`, network.py:147-162: the ReLU mask and the bias
// gradient both want a pass over dy): g = y > 0 ? dy : 0 written, sum(g) per (channel, slice) in double
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            float* __restrict__ gout, int N, int C, int S, int slices,
                                                            int vec, double* __restrict__ part) {
    __shared__ double red[4];
    const int c = blockIdx.x, sl = blockIdx.y;
    const int total = N * S;
    int len = (total + slices - 1) / slices;
    if (vec) len = (len + 3) & ~3;
    const int b = min(total, sl * len), e = min(total, b + len);
    float s1 = 0.f;
    if (vec) {
        for (int i = b + 4 * (int)threadIdx.x; i < e; i += 1024) {
            const int n = i / S, s = i - n * S;
            const size_t at = ((size_t)n * C + c) * S + s;
            const float4 d = *reinterpret_cast<const float4*>(dy + at), v = *reinterpret_cast<const float4*>(y + at);
            const float4 r = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
            *reinterpret_cast<float4*>(gout + at) = r;
            s1 += (r.x + r.y) + (r.z + r.w);
        }
    } else {
        for (int i = b + (int)threadIdx.x; i < e; i += 256) {
            const int n = i / S, s = i - n * S;
            const size_t at = ((size_t)n * C + c) * S + s;
            const float r = y[at] > 0.f ? dy[at] : 0.f;
            gout[at] = r;
            s1 += r;
        }
    }
    const double t = block_sum_256<double>((double)s1, red);
    if (threadIdx.x == 0) part[(size_t)c * slices + sl] = t;
}

__global__ void channel_sum_final_kernel(const double* __restrict__ part, int C, int slices, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < slices; ++k) s += part[(size_t)c * slices + k];
    out[c] = (float)s;
}

// torch.optim.Adam (no amsgrad, no weight decay), single-tensor formulation:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float b1,
                                                   float b2, float eps, float step_size, float inv_sqrt_bc2) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] -= step_size * (mi / denom);
    }
}

}  // namespace zsv

using namespace zsv;

extern "C" int zsv_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y) return ZSV_E_NULL;
    if (n <= 0) return n == 0 ? ZSV_OK : ZSV_E_BAD_SHAPE;
    const int vec = aligned16(x) && aligned16(y);
    hipLaunchKernelGGL((ew_kernel<0>), dim3(ew_blocks(vec ? n / 4 : n)), dim3(256), 0, (hipStream_t)stream, x, nullptr, y, (long)n, vec);
    return launch_status();
}

extern "C" int zsv_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    if (!dy || !y || !dx) return ZSV_E_NULL;
    if (n <= 0) return n == 0 ? ZSV_OK : ZSV_E_BAD_SHAPE;
    const int vec = aligned16(dy) && aligned16(y) && aligned16(dx);
    hipLaunchKernelGGL((ew_kernel<1>), dim3(ew_blocks(vec ? n / 4 : n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (long)n, vec);
    return launch_status();
}

extern "C" int zsv_add_relu_fwd(const float* a, const float* b, float* y, int64_t n, void* stream) {
    if (!a || !b || !y) return ZSV_E_NULL;
    if (n <= 0) return n == 0 ? ZSV_OK : ZSV_E_BAD_SHAPE;
    const int vec = aligned16(a) && aligned16(b) && aligned16(y);
    hipLaunchKernelGGL((ew_kernel<2>), dim3(ew_blocks(vec ? n / 4 : n)), dim3(256), 0, (hipStream_t)stream, a, b, y, (long)n, vec);
    return launch_status();
}

extern "C" int zsv_meanpool_fwd(const float* x, int32_t N, int32_t C, int32_t S, float* y, void* stream) {
    if (!x || !y) return ZSV_E_NULL;
    if (N <= 0 || C <= 0 || S <= 0) return ZSV_E_BAD_SHAPE;
    const int rows = N * C;
    hipLaunchKernelGGL(meanpool_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows, S, y);
    return launch_status();
}

extern "C" int zsv_meanpool_bwd(const float* dy, int32_t N, int32_t C, int32_t S, float* dx, void* stream) {
    if (!dy || !dx) return ZSV_E_NULL;
    if (N <= 0 || C <= 0 || S <= 0) return ZSV_E_BAD_SHAPE;
    const long total = (long)N * C * S;
    hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, total, S, dx);
    return launch_status();
}

static int pool_geom(PoolGeom& g, int32_t Ti, int32_t Hi, int32_t Wi, int32_t kT, int32_t kH, int32_t kW, int32_t pT,
                     int32_t pH, int32_t pW, int32_t To, int32_t Ho, int32_t Wo) {
    if (Ti <= 0 || Hi <= 0 || Wi <= 0 || kT <= 0 || kH <= 0 || kW <= 0 || pT < 0 || pH < 0 || pW < 0) return ZSV_E_BAD_SHAPE;
    if (2 * pT > kT || 2 * pH > kH || 2 * pW > kW) return ZSV_E_BAD_SHAPE;   // torch: pad <= kernel/2
    // floor mode, stride == kernel
    if (To != (Ti + 2 * pT - kT) / kT + 1 || Ho != (Hi + 2 * pH - kH) / kH + 1 || Wo != (Wi + 2 * pW - kW) / kW + 1)
        return ZSV_E_BAD_SHAPE;
    g = PoolGeom{Ti, Hi, Wi, To, Ho, Wo, kT, kH, kW, pT, pH, pW};
    return ZSV_OK;
}

extern "C" int zsv_maxpool3d_fwd(const float* x, int32_t N, int32_t C, int32_t Ti, int32_t Hi, int32_t Wi, int32_t kT,
                                 int32_t kH, int32_t kW, int32_t pT, int32_t pH, int32_t pW, int32_t To, int32_t Ho,
                                 int32_t Wo, float* y, int32_t* argmax, void* stream) {
    if (!x || !y || !argmax) return ZSV_E_NULL;
    PoolGeom g;
    int st = pool_geom(g, Ti, Hi, Wi, kT, kH, kW, pT, pH, pW, To, Ho, Wo);
    if (st) return st;
    if (N <= 0 || C <= 0) return ZSV_E_BAD_SHAPE;
    const long total = (long)N * C * To * Ho * Wo;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, total, g, y, argmax);
    return launch_status();
}

extern "C" int zsv_maxpool3d_bwd(const float* dy, const int32_t* argmax, int32_t N, int32_t C, int32_t Ti, int32_t Hi,
                                 int32_t Wi, int32_t kT, int32_t kH, int32_t kW, int32_t pT, int32_t pH, int32_t pW,
                                 int32_t To, int32_t Ho, int32_t Wo, float* dx, void* stream) {
    if (!dy || !argmax || !dx) return ZSV_E_NULL;
    PoolGeom g;
    int st = pool_geom(g, Ti, Hi, Wi, kT, kH, kW, pT, pH, pW, To, Ho, Wo);
    if (st) return st;
    if (N <= 0 || C <= 0) return ZSV_E_BAD_SHAPE;
    const long total = (long)N * C * Ti * Hi * Wi;
    const long iS = (long)Ti * Hi * Wi;
    if (Wi % 4 == 0 && aligned16(dx) && iS < (1L << 30)) {
        const long rows = (long)N * C, quads = iS / 4;
        long bx = (quads + 255) / 256;
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(maxpool_bwd_quad_kernel, dim3((unsigned)bx, (unsigned)(rows > 32768 ? 32768 : rows)), dim3(256), 0,
                           (hipStream_t)stream, dy, argmax, (int)rows, g, dx);
        return launch_status();
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, argmax, total, g, dx);
    return launch_status();
}

extern "C" size_t zsv_channel_sum_workspace_bytes(int32_t N, int32_t C, int32_t S) {
    if (N <= 0 || C <= 0 || S <= 0) return 0;
    return (size_t)C * cs_slices(N, C, S) * sizeof(double);
}

extern "C" int zsv_channel_sum(const float* dy, int32_t N, int32_t C, int32_t S, float* db, void* workspace,
                               size_t workspace_bytes, void* stream_) {
    if (!dy || !db || !workspace) return ZSV_E_NULL;
    if (N <= 0 || C <= 0 || S <= 0) return ZSV_E_BAD_SHAPE;
    if ((double)N * C * S >= 2147483647.0) return ZSV_E_TOO_LARGE;
    if (workspace_bytes < zsv_channel_sum_workspace_bytes(N, C, S)) return ZSV_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int slices = cs_slices(N, C, S);
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, slices), dim3(256), 0, stream, dy, N, C, S, slices, (double*)workspace);
    int st = launch_status();
    if (st) return st;
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 127) / 128), dim3(128), 0, stream, (const double*)workspace, C, slices, db);
    return launch_status();
}

extern "C" int zsv_relu_bwd_bias(const float* dy, const float* y, float* g, int32_t N, int32_t C, int32_t S, float* db,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
    if (!dy || !y || !g || !db || !workspace) return ZSV_E_NULL;
    if (N <= 0 || C <= 0 || S <= 0) return ZSV_E_BAD_SHAPE;
    if ((double)N * C * S >= 2147483647.0) return ZSV_E_TOO_LARGE;
    if (workspace_bytes < zsv_channel_sum_workspace_bytes(N, C, S)) return ZSV_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int slices = cs_slices(N, C, S);
    // float4 path: whole 16-byte pieces per (n, c) row AND 16-byte aligned bases (a contiguous view of a larger buffer may
    // start at an odd offset)
    const int vec = (S % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(g)) ? 1 : 0;
    hipLaunchKernelGGL(relu_bwd_bias_kernel, dim3(C, slices), dim3(256), 0, stream, dy, y, g, N, C, S, slices, vec, (double*)workspace);
    int st = launch_status();
    if (st) return st;
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 127) / 128), dim3(128), 0, stream, (const double*)workspace, C, slices, db);
    return launch_status();
}

extern "C" int zsv_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int32_t step, void* stream) {
    if (!p || !g || !exp_avg || !exp_avg_sq) return ZSV_E_NULL;
    if (n <= 0 || step <= 0) return n == 0 ? ZSV_OK : ZSV_E_BAD_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, (long)n,
                       beta1, beta2, eps, step_size, inv_sqrt_bc2);
    return launch_status();
}

// ---- multi-tensor Adam: one launch for every parameter tensor of the model ---------------------
// table[i] = {p, g, exp_avg, exp_avg_sq, n, first_chunk}; a chunk is 256 threads x 16 elements.
namespace zsv {
constexpr int ADAM_CHUNK = 4096;
__global__ __launch_bounds__(256) void adam_multi_kernel(const zsv_adam_tensor* __restrict__ table, int count, float b1,
                                                         float b2, float eps, float step_size, float inv_sqrt_bc2) {
    // binary search: last tensor whose first_chunk <= blockIdx.x
    int lo = 0, hi = count - 1;
    const long chunk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
    }
    const zsv_adam_tensor t = table[lo];
    const long base = (chunk - t.first_chunk) * ADAM_CHUNK;
    const long end = min(t.n, base + ADAM_CHUNK);
    for (long i = base + threadIdx.x; i < end; i += 256) {
        const float gi = t.g[i];
        const float mi = b1 * t.exp_avg[i] + (1.f - b1) * gi;
        const float vi = b2 * t.exp_avg_sq[i] + (1.f - b2) * gi * gi;
        t.exp_avg[i] = mi;
        t.exp_avg_sq[i] = vi;
        t.p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}
}  // namespace zsv

extern "C" int zsv_adam_multi(const zsv_adam_tensor* table_device, int32_t count, int64_t total_chunks, float lr,
                              float beta1, float beta2, float eps, int32_t step, void* stream) {
    if (!table_device) return ZSV_E_NULL;
    if (count <= 0 || total_chunks <= 0 || step <= 0 || total_chunks > 0x7fffffffL) return count == 0 ? ZSV_OK : ZSV_E_BAD_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(zsv::adam_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, table_device,
                       count, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
    return launch_status();
}

// ---- loss scaling (torch.cuda.amp.GradScaler, main.py:137,195-203) on the device -----------------------
// scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update() without a host round trip:
//   grad_check_multi : found_inf |= any non-finite scaled gradient   (_amp_foreach_non_finite_check_and_unscale_)
//   adam_multi_scaled: skipped when found_inf; gradients multiplied by 1/scale in registers; the bias correction
//                      uses the device-side count of steps actually taken
//   scaler_update    : _amp_update_scale_ (backoff / growth) + steps_done += !found_inf, found_inf = 0
namespace zsv {
__global__ __launch_bounds__(256) void grad_check_multi_kernel(const zsv_adam_tensor* __restrict__ table, int count,
                                                               zsv_scaler_state* __restrict__ st) {
    int lo = 0, hi = count - 1;
    const long chunk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
    }
    const zsv_adam_tensor t = table[lo];
    const long base = (chunk - t.first_chunk) * ADAM_CHUNK;
    const long end = min(t.n, base + ADAM_CHUNK);
    bool bad = false;
    for (long i = base + threadIdx.x; i < end; i += 256) {
        const float g = t.g[i];
        bad |= !(fabsf(g) <= 3.402823466e38f);                    // inf or NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) st->found_inf = 1;   // benign race: every writer stores 1
}

__global__ __launch_bounds__(256) void adam_multi_scaled_kernel(const zsv_adam_tensor* __restrict__ table, int count, float lr,
                                                                float b1, float b2, float eps,
                                                                const zsv_scaler_state* __restrict__ st) {
    if (st->found_inf) return;                                     // scaler.step skips optimizer.step (main.py:200)
    const float inv_scale = (float)(1.0 / (double)st->scale);      // as GradScaler: scale.double().reciprocal().float()
    const int step = st->steps_done + 1;
    const float step_size = (float)((double)lr / (1.0 - pow((double)b1, (double)step)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)step)));
    int lo = 0, hi = count - 1;
    const long chunk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
    }
    const zsv_adam_tensor t = table[lo];
    const long base = (chunk - t.first_chunk) * ADAM_CHUNK;
    const long end = min(t.n, base + ADAM_CHUNK);
    for (long i = base + threadIdx.x; i < end; i += 256) {
        const float gi = t.g[i] * inv_scale;
        const float mi = b1 * t.exp_avg[i] + (1.f - b1) * gi;
        const float vi = b2 * t.exp_avg_sq[i] + (1.f - b2) * gi * gi;
        t.exp_avg[i] = mi;
        t.exp_avg_sq[i] = vi;
        t.p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

__global__ void scaler_update_kernel(zsv_scaler_state* st, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->found_inf) {
        st->scale *= backoff;
        st->growth_tracker = 0;
    } else {
        const int ok = st->growth_tracker + 1;
        if (ok == interval) {
            const float grown = st->scale * growth;
            if (fabsf(grown) <= 3.402823466e38f) st->scale = grown;   // torch keeps the old scale if growth overflows
            st->growth_tracker = 0;
        } else {
            st->growth_tracker = ok;
        }
        st->steps_done += 1;
    }
    st->found_inf = 0;
}
}  // namespace zsv

extern "C" int zsv_grad_check_multi(const zsv_adam_tensor* table_device, int32_t count, int64_t total_chunks,
                                    zsv_scaler_state* state_device, void* stream) {
    if (!table_device || !state_device) return ZSV_E_NULL;
    if (count <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffL) return count == 0 ? ZSV_OK : ZSV_E_BAD_SHAPE;
    hipLaunchKernelGGL(zsv::grad_check_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, table_device,
                       count, state_device);
    return launch_status();
}

extern "C" int zsv_adam_multi_scaled(const zsv_adam_tensor* table_device, int32_t count, int64_t total_chunks, float lr,
                                     float beta1, float beta2, float eps, const zsv_scaler_state* state_device, void* stream) {
    if (!table_device || !state_device) return ZSV_E_NULL;
    if (count <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffL) return count == 0 ? ZSV_OK : ZSV_E_BAD_SHAPE;
    hipLaunchKernelGGL(zsv::adam_multi_scaled_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, table_device,
                       count, lr, beta1, beta2, eps, state_device);
    return launch_status();
}

extern "C" int zsv_scaler_update(zsv_scaler_state* state_device, float growth_factor, float backoff_factor,
                                 int32_t growth_interval, void* stream) {
    if (!state_device) return ZSV_E_NULL;
    if (!(growth_factor > 1.f) || !(backoff_factor > 0.f && backoff_factor < 1.f) || growth_interval <= 0) return ZSV_E_BAD_SHAPE;
    hipLaunchKernelGGL(zsv::scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state_device, growth_factor,
                       backoff_factor, growth_interval);
    return launch_status();
}

extern "C" const char* zsv_status_string(int status) {
    switch (status) {
        case ZSV_OK: return "ok";
        case ZSV_E_BAD_SHAPE: return "bad shape / inconsistent geometry";
        case ZSV_E_NULL: return "required pointer is NULL";
        case ZSV_E_WORKSPACE: return "workspace too small";
        case ZSV_E_TOO_LARGE: return "tensor has >= 2^31 elements";
        case ZSV_E_LAUNCH: return "kernel launch failed";
        case ZSV_E_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}

extern "C" const char* zsv_version(void) { return "zsv_hip gfx950 " __DATE__; }
