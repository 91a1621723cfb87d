// umpa_tiled.h -- tiled fast path (placeholder until the kernels land).
#pragma once
#include "umpa_direct.h"
namespace umpa {
struct TiledState {};
struct TiledTimers { int n = 0; int name[8]; hipEvent_t t0[8], t1[8]; };
inline bool tiled_supported(int, int, int) { return false; }
inline int tiled_match(TiledState&, const ModelDev&, int, int, int, const RegionArgs&, hipStream_t, TiledTimers*) { return -1; }
inline void tiled_release(TiledState&) {}
}
