// knobs.hip -- snapshot of the ZSV_* environment switches (knobs.h).
#include <stdlib.h>
#include <string.h>
#include "zsv_hip.h"
#include "knobs.h"

namespace zsv {

const char* volatile g_knobs[K_COUNT];

namespace {
const char* const kNames[K_COUNT] = {
#define ZSV_KNOB_NAME(name) "ZSV_" #name,
    ZSV_KNOB_LIST(ZSV_KNOB_NAME)
#undef ZSV_KNOB_NAME
};

// Values are copied (the environment block may be rewritten later); an older copy is never freed, so a launch that
// still holds its pointer stays valid (a reload happens a handful of times per process, in tests and A/B tools).
int snapshot() {
    int set = 0;
    for (int i = 0; i < K_COUNT; ++i) {
        const char* v = getenv(kNames[i]);
        const char* old = g_knobs[i];
        if (v == nullptr) { g_knobs[i] = nullptr; continue; }
        ++set;
        if (old != nullptr && strcmp(old, v) == 0) continue;
        g_knobs[i] = strdup(v);
    }
    return set;
}

const int g_loaded = snapshot();       // at library load
}  // namespace

}  // namespace zsv

extern "C" int32_t zsv_reload_knobs(void) { return (int32_t)zsv::snapshot(); }
