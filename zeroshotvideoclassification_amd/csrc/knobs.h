// knobs.h -- the debug / A-B environment switches of the library, read ONCE.
//
// Every `ZSV_*` variable the launch path looks at is listed here.  The values are snapshotted when the library is
// loaded and again on `zsv_reload_knobs()`; the launch path reads the snapshot (an array load) instead of calling
// getenv() -- which walks the whole environment and is not safe against a concurrent setenv() -- from the autograd
// thread on every convolution call.  `ZSV_KNOB(NAME)` yields the value of ZSV_NAME (const char*, nullptr when unset).
#pragma once

#define ZSV_KNOB_LIST(X) \
    X(NO_FUSED_STATS) \
    X(NO_BN_FUSION) \
    X(BF16_NO_SAME) \
    X(BF16_NO_TSAME) \
    X(BF16_NO_SAME64) \
    X(BF16_NO_SAME9) \
    X(BF16_SAME9_MIN_CHUNKS) \
    X(WINOT_NO_T32) \
    X(BF16_GROUP_OUTER) \
    X(BF16_NO_WGRAD) \
    X(BF16_NO_WGRAD_GATHER) \
    X(BF16_WGRAD_WGS) \
    X(NO_DGRAD_S2) \
    X(NO_DGRAD_S2T) \
    X(DGRAD_S2_BN) \
    X(DGRAD_S2_KS) \
    X(NO_DOWN_FUSION) \
    X(DGRAD_S2_NO_X4) \
    X(CONV_CFG) \
    X(NO_SPLITK) \
    X(TAP_KS) \
    X(SPLITK_THRESH) \
    X(NO_CFG5) \
    X(NO_TAP) \
    X(NO_LDS_EPILOGUE) \
    X(WGRAD_CFG) \
    X(WGRAD_BN) \
    X(WGRAD_WGS) \
    X(WGRAD_NO_TWOTAP) \
    X(NO_WGRAD_DMA) \
    X(WGRAD_DMA_TM) \
    X(WGRAD_DMA_TN) \
    X(WGRAD_DMA_SLICES) \
    X(WGRAD_LINEAR_WALK) \
    X(NO_STEM_WGRAD) \
    X(STEM_WGRAD_WGS) \
    X(WGRAD_TRING_RESIDENT) \
    X(WGRAD_TRING_SLICES) \
    X(NO_WGRAD_TRING) \
    X(NO_WGRAD_TWINO) \
    X(WGRAD_WINO_SLICES) \
    X(NO_WINO) \
    X(NO_WGRAD_WINO) \
    X(NO_WGRAD_WINO4) \
    X(WINO_MIN_TILES) \
    X(WINO_NO_SPLITK) \
    X(NO_WINOT) \
    X(WINOT_MIN_TILES) \
    X(BN_NT_MB) \
    X(BN_NO_MASKED_G) \
    X(WINOT_MAX_WASTE) \
    X(NO_WINO_FWD) \
    X(WINO_NO_F43) \
    X(WINOT_NO_F43) \
    X(WINO_NO_X4) \
    X(NO_SLAB_SUM_ROWS) \
    X(SPLITK_REDUCE_SCALAR) \
    X(NO_PACK_TILED) \
    X(NO_T2_DENSE) \
    X(WINO_NO_VW) \
    X(WINO_VW_MIN_WGS) \
    X(WINOT_GENERIC_EPILOGUE) \
    X(WINO_GENERIC_EPILOGUE)

namespace zsv {
enum KnobId {
#define ZSV_KNOB_ENUM(name) K_##name,
    ZSV_KNOB_LIST(ZSV_KNOB_ENUM)
#undef ZSV_KNOB_ENUM
    K_COUNT
};
extern const char* volatile g_knobs[K_COUNT];
}  // namespace zsv

#define ZSV_KNOB(name) (::zsv::g_knobs[::zsv::K_##name])
