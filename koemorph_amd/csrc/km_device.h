// Device helpers shared by km_mel.hip and km_core.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "koemorph.h"

namespace km {

struct LogParams {
    int log_mode;
    float amin, top_db, db_add, db_scale, log_eps;
};

// per-window constants of librosa.power_to_db(ref=np.max): reference level and the top_db floor
__device__ __forceinline__ void log_window_consts(const LogParams& p, float ref, float& ref_db, float& floor_db) {
    ref_db = 10.0f * __log10f(fmaxf(p.amin, ref));
    // log_spec.max() - top_db: the window maximum is its own reference, so max_db = f(ref) - ref_db
    floor_db = (10.0f * __log10f(fmaxf(p.amin, ref)) - ref_db) - p.top_db;
}

__device__ __forceinline__ float log_one(const LogParams& p, float s, float ref_db, float floor_db) {
    // hardware v_log_f32 (1 ulp in log2): |error| < 2e-5 dB over the whole [amin, 1e10] range; the
    // operands are clamped to >= amin / eps, so no denormal reaches it
    if (p.log_mode == KM_LOG_LN_EPS) return __logf(s + p.log_eps);    // src/features/stft.py:123
    float v = 10.0f * __log10f(fmaxf(p.amin, s)) - ref_db;            // librosa.power_to_db
    v = fmaxf(v, floor_db);
    return (v + p.db_add) * p.db_scale;                               // simplified_dual_stream_model.py:200
}

}  // namespace km
