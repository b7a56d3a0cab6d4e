// nearest_class.hip -- cosine nearest classes in the 300-d embedding space (gfx950).
//
// Replaces, on the device, `cdist(embed, class_embed, 'cosine').argsort(1)[:, :k]` of the
// reference's compute_accuracy (main.py:316-325) and the per-step train accuracy
// (main.py:182-185).  Two launches:
//
//   cosine_dist_kernel : D[r][c] = 1 - <p_r, e_c> / (|p_r| |e_c|) in DOUBLE precision on the matrix
//                        core (v_mfma_f64_16x16x4_f64; the fp32 inputs convert exactly, so the
//                        distances agree with scipy's double cdist to ~1e-16 and the ranking is the
//                        same down to genuine ties);
//   row_topk_kernel    : per row the k smallest (distance, class index) pairs in lexicographic
//                        order -- what a stable argsort returns.
//
// The problem is tiny (rows x classes x 300: UCF101 / HMDB51 / ActivityNet tables of 101 / 51 / 200
// classes), so the kernels are written for exactness, not for a roofline.
#include "zsv_common.h"
#include "zsv_hip.h"

namespace zsv {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// One wave = one 16-row block of `pred` against every 16-class block of `classes`.
// A fragment: lane holds A[m = lane & 15][k = lane >> 4]; B fragment: B[k = lane >> 4][n = lane & 15];
// C/D (f64 form): col = lane & 15, row = (lane >> 4) + 4 * reg   (cdna_hip_programming.md, MFMA maps).
__global__ __launch_bounds__(256) void cosine_dist_kernel(const float* __restrict__ pred, const float* __restrict__ classes,
                                                          int rows, int dim, int n_classes, int cpad,
                                                          double* __restrict__ dist) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x * 4 + wave;                      // 16-row block of this wave
    if (rb * 16 >= rows) return;
    const int m = lane & 15, kq = lane >> 4;
    const int row_a = min(rb * 16 + m, rows - 1);              // clamped: the extra rows are never stored
    const float* pa = pred + (size_t)row_a * dim;
    const int ksteps = (dim + 3) >> 2;

    // |p_r|^2 for the row this lane feeds (partial over k = kq mod 4, then over the 4 lane groups)
    double na = 0.0;
    for (int s = 0; s < ksteps; ++s) {
        const int k = 4 * s + kq;
        const double a = k < dim ? (double)pa[k] : 0.0;
        na += a * a;
    }
    na += __shfl_xor(na, 16, 64);
    na += __shfl_xor(na, 32, 64);                              // every lane: |row (lane & 15)|^2
    double nrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) nrow[i] = sqrt(__shfl(na, kq + 4 * i, 64));

    for (int cb = 0; cb * 16 < n_classes; ++cb) {
        const int cls = min(cb * 16 + m, n_classes - 1);
        const float* pb = classes + (size_t)cls * dim;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        double nb = 0.0;
        for (int s = 0; s < ksteps; ++s) {
            const int k = 4 * s + kq;
            const double a = k < dim ? (double)pa[k] : 0.0;
            const double b = k < dim ? (double)pb[k] : 0.0;
            nb += b * b;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        nb += __shfl_xor(nb, 16, 64);
        nb += __shfl_xor(nb, 32, 64);                          // |class (lane & 15) of this block|^2
        const double ncol = sqrt(nb);
        const int c = cb * 16 + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rb * 16 + kq + 4 * i;
            if (r < rows && c < n_classes) {
                double cosine = acc[i] / (nrow[i] * ncol);
                if (fabs(cosine) > 1.0) cosine = copysign(1.0, cosine);      // scipy clips rounding error the same way
                dist[(size_t)r * cpad + c] = 1.0 - cosine;
            }
        }
    }
}

// One wave per row: k rounds of "smallest (d, c) greater than the last one taken".
__global__ __launch_bounds__(256) void row_topk_kernel(const double* __restrict__ dist, int rows, int n_classes, int cpad,
                                                       int k, int* __restrict__ out_index, double* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const double* d = dist + (size_t)r * cpad;
    const double inf = __builtin_huge_val();
    double last_d = -inf;
    int last_c = -1;
    for (int j = 0; j < k; ++j) {
        double best_d = inf;
        int best_c = 0x7fffffff;
        for (int c = lane; c < n_classes; c += 64) {
            double v = d[c];
            if (v != v) v = inf;                                // NaN (zero-norm row) ranks last
            const bool after = v > last_d || (v == last_d && c > last_c);
            const bool better = v < best_d || (v == best_d && c < best_c);
            if (after && better) { best_d = v; best_c = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_xor(best_d, o, 64);
            const int oc = __shfl_xor(best_c, o, 64);
            if (od < best_d || (od == best_d && oc < best_c)) { best_d = od; best_c = oc; }
        }
        last_d = best_d;
        last_c = best_c;
        if (lane == 0) {
            const bool found = best_c != 0x7fffffff;
            out_index[(size_t)r * k + j] = found ? best_c : -1;
            if (out_dist) out_dist[(size_t)r * k + j] = found ? d[best_c] : inf;
        }
    }
}

}  // namespace zsv

static inline int32_t class_pitch(int32_t n_classes) { return (n_classes + 15) & ~15; }

extern "C" size_t zsv_cosine_topk_workspace_bytes(int32_t rows, int32_t n_classes) {
    if (rows <= 0 || n_classes <= 0) return 0;
    return (size_t)rows * (size_t)class_pitch(n_classes) * sizeof(double);
}

extern "C" int zsv_cosine_topk(const float* embed, const float* class_embed, int32_t rows, int32_t dim, int32_t n_classes,
                               int32_t k, int32_t* out_index, double* out_dist, void* workspace, size_t workspace_bytes,
                               void* stream_) {
    if (!embed || !class_embed || !out_index || !workspace) return ZSV_E_NULL;
    if (rows <= 0 || dim <= 0 || n_classes <= 0 || k <= 0 || k > n_classes) return ZSV_E_BAD_SHAPE;
    if ((double)rows * dim >= 2147483647.0 || (double)n_classes * dim >= 2147483647.0) return ZSV_E_TOO_LARGE;
    if (workspace_bytes < zsv_cosine_topk_workspace_bytes(rows, n_classes)) return ZSV_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int cpad = class_pitch(n_classes);
    const int row_blocks = (rows + 15) / 16;
    hipLaunchKernelGGL(zsv::cosine_dist_kernel, dim3((row_blocks + 3) / 4), dim3(256), 0, stream, embed, class_embed, rows, dim,
                       n_classes, cpad, (double*)workspace);
    int st = zsv::launch_status();
    if (st) return st;
    hipLaunchKernelGGL(zsv::row_topk_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const double*)workspace, rows,
                       n_classes, cpad, k, out_index, out_dist);
    return zsv::launch_status();
}
