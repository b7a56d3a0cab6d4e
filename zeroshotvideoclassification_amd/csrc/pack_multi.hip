// pack_multi.hip -- every weight panel of a training step in ONE launch (zsv_hip.h: zsv_pack_multi), and the thread-local
// context through which the *_panel entry points reach the pack sites of the convolution paths.
//
// A job is a pack launch written down by the path that would have made it (PANEL_RECORD, conv_params.h): kernel kind, scalar
// arguments, the weight tensor and the panel.  Block b of the grid serves 1024 consecutive panel elements of the job whose
// [first_block, next first_block) range holds b (binary search over the table): the ~74 panels of R(2+1)D-18 -- forward and
// input-gradient forms of its 37 convolutions, 0.8 GB written -- are filled by one grid instead of 76 launches that each
// under-fill the chip.  The element formulas are the per-call kernels' own (pack_bodies.h): bit-identical panels.
#include "pack_bodies.h"

namespace zsv {

thread_local PanelCtx g_panel = {PANEL_NONE, nullptr, 0, 0, 0, {}};

__global__ __launch_bounds__(256) void pack_multi_kernel(const zsv_pack_job* __restrict__ jobs, int count) {
    const long b = blockIdx.x;
    int lo = 0, hi = count - 1;                          // last job with first_block <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const zsv_pack_job& J = jobs[lo];
    const long i0 = (b - J.first_block) * 1024;
    const long i1 = i0 + 1024 < J.total ? i0 + 1024 : J.total;
    const float* __restrict__ W = J.w;
    float* __restrict__ out = J.out;
    if (J.kind == 0) {
        const PackTapArgs a = {J.i[0], J.i[1], J.i[2], J.i[3], J.i[4], J.i[5], J.i[6], J.i[7], J.i[8], J.i[9], J.i[10], J.i[11], J.i[12],
                               J.i[13], J.i[14], J.i[15], J.i[16], J.i[17], J.i[18]};
        for (long i = i0 + threadIdx.x; i < i1; i += 256) out[i] = pack_tap_value(a, W, i);
    } else if (J.kind == 1 || J.kind == 2) {
        // per (channel, row, row tap) triple: three tap reads, NP stores (the job's blocks share its triples evenly)
        const PackWinoArgs a = {J.i[0], J.i[1], J.i[2], J.i[3], J.i[4], J.i[5], (long)J.l[0], (long)J.l[1]};
        const int np = J.kind == 1 ? 4 : 6;
        const long triples = J.total / np, nblocks = (J.total + 1023) / 1024;
        const long per = (triples + nblocks - 1) / nblocks;
        const long t0 = (b - J.first_block) * per, t1 = t0 + per < triples ? t0 + per : triples;
        if (J.kind == 1) { for (long t = t0 + threadIdx.x; t < t1; t += 256) pack_wino_triple<4>(a, W, out, t); }
        else { for (long t = t0 + threadIdx.x; t < t1; t += 256) pack_wino_triple<6>(a, W, out, t); }
    } else {
        const PackS2Args a = {J.i[0], J.i[1], J.i[2], J.i[3]};
        for (long i = i0 + threadIdx.x; i < i1; i += 256) out[i] = pack_s2_value(a, W, i);
    }
}

}  // namespace zsv

using namespace zsv;

extern "C" int zsv_pack_multi(const zsv_pack_job* jobs_device, int32_t count, int64_t total_blocks, void* stream) {
    if (!jobs_device) return ZSV_E_NULL;
    if (count <= 0 || total_blocks <= 0) return (count == 0 && total_blocks == 0) ? ZSV_OK : ZSV_E_BAD_SHAPE;
    if (total_blocks > 0x7fffffffL) return ZSV_E_TOO_LARGE;
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, (int)count);
    return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
}
