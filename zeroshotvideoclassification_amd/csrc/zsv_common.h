// zsv_common.h -- small device/host helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zsv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 64-lane wave reductions (wave = 64 on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread workgroup; result valid in every thread. `scratch` holds >= 4 T.
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// unsigned division by a run-time constant without the ~25-instruction divide: q = (umulhi(p, mul) + p) >> shift, exact for p < 2^31
struct Magic { unsigned mul, shift; };
static inline Magic make_magic(unsigned d) {
    Magic m;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    m.shift = s;
    m.mul = (unsigned)((((1ull << s) - d) << 32) / d + 1);     // ceil(2^(32+s)/d) - 2^32
    return m;
}
__device__ __forceinline__ unsigned mdiv(unsigned p, Magic m) { return (__umulhi(p, m.mul) + p) >> m.shift; }

static inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 5; }

}  // namespace zsv
