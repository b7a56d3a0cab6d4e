// preprocess.hip -- clip pre-processing on the GPU (gfx950): u8 frames -> model input.
//
// Fuses the reference's per-clip transform chain (auxiliary/transforms.py:41-56) into one
// HBM pass:  ToFloatTensorInZeroOne ((u8/255 - 1)/2 and THWC -> CTHW, transforms.py:116-117)
//            -> Resize(short side 128, bilinear, align_corners=False, transforms.py:99-107)
//            -> Center/RandomCrop(112) (transforms.py:80-97,132-158)
//            -> RandomHorizontalFlip (transforms.py:188-195)
// The CPU pipeline materialises the resized 3xTx128xW' float tensor per clip; here every
// output voxel reads its 4 source pixels directly (u8, 1 B each) -- 13 MB of u8 in, 53 MB of
// fp32 out per 22-clip batch instead of an fp32 host->device copy.  HBM-bound; lanes walk the
// output W axis (coalesced 4-B stores; the u8 gathers of a row hit the same cache lines).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zsv_hip.h"
#include "zsv_common.h"

namespace zsv {

struct ClipGeom {
    int T, Hin, Win;         // source frames (T, Hin, Win, 3) u8, HWC interleaved
    int Hres, Wres;          // size after the resize
    int crop;                // output is crop x crop
    float inv_scale;         // source step per resized pixel (1 / scale_factor)
};

__device__ __forceinline__ float norm_u8(uint8_t v) { return ((float)v / 255.f - 1.0f) / 2.0f; }

__global__ __launch_bounds__(256) void clip_transform_kernel(const uint8_t* __restrict__ frames, const int* __restrict__ params,
                                                             ClipGeom g, long total, float* __restrict__ out) {
    const int cc = g.crop * g.crop;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int x = (int)(i % g.crop);
        long r = i / g.crop;
        const int y = (int)(r % g.crop);
        r /= g.crop;
        const int t = (int)(r % g.T);
        r /= g.T;
        const int c = (int)(r % 3);
        const int n = (int)(r / 3);
        const int top = params[3 * n + 0], left = params[3 * n + 1], flip = params[3 * n + 2];
        if (flip) x = g.crop - 1 - x;                       // flip acts on the cropped clip
        const int ry = top + y, rx = left + x;              // pixel of the resized frame
        // torch upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5, clamped at 0
        float sy = g.inv_scale * ((float)ry + 0.5f) - 0.5f;
        float sx = g.inv_scale * ((float)rx + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = min((int)sy, g.Hin - 1), x0 = min((int)sx, g.Win - 1);
        const int y1 = min(y0 + 1, g.Hin - 1), x1 = min(x0 + 1, g.Win - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const uint8_t* f = frames + (((size_t)n * g.T + t) * g.Hin) * g.Win * 3 + c;
        const float v00 = norm_u8(f[((size_t)y0 * g.Win + x0) * 3]), v01 = norm_u8(f[((size_t)y0 * g.Win + x1) * 3]);
        const float v10 = norm_u8(f[((size_t)y1 * g.Win + x0) * 3]), v11 = norm_u8(f[((size_t)y1 * g.Win + x1) * 3]);
        const float top_row = (1.f - lx) * v00 + lx * v01;
        const float bot_row = (1.f - lx) * v10 + lx * v11;
        out[i] = (1.f - ly) * top_row + ly * bot_row;
        (void)cc;
    }
}

}  // namespace zsv

using namespace zsv;

extern "C" int zsv_clip_transform(const uint8_t* frames_u8, int32_t N, int32_t T, int32_t Hin, int32_t Win,
                                  int32_t Hres, int32_t Wres, float inv_scale, int32_t crop,
                                  const int32_t* crop_flip_params_device, float* out, void* stream) {
    if (!frames_u8 || !crop_flip_params_device || !out) return ZSV_E_NULL;
    if (N <= 0 || T <= 0 || Hin <= 0 || Win <= 0 || crop <= 0 || Hres < crop || Wres < crop || !(inv_scale > 0.f))
        return ZSV_E_BAD_SHAPE;
    const long total = (long)N * 3 * T * crop * crop;
    if ((double)total >= 2147483647.0 * 4 || (double)N * T * Hin * Win * 3 >= 4.0e9) return ZSV_E_TOO_LARGE;
    ClipGeom g = {T, Hin, Win, Hres, Wres, crop, inv_scale};
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(clip_transform_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames_u8,
                       (const int*)crop_flip_params_device, g, total, out);
    return launch_status();
}
