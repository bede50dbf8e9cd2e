// Output tail of the legacy KoeMorphModel (decoder.py:162-177, :260-340, :384-456), shared by kmm_tail_kernel (km_koemorph.hip,
// the launch-per-step chain) and the fused decode kernel (km_kmmf.hip).  Included inside namespace km of a .hip file.
#pragma once

#include <cmath>
#include <cstdint>

// output activation, mix with the previous frame, temporal smoothing with the caller's state, constraints
struct KmmTail {
    const float* h; int hid, NB;
    const float* wout; const float* bout; const float* prev;
    int out_act;                 // 0 sigmoid, 1 tanh, 2 none (decoder.py:162-167)
    int smooth;                  // -1 off, 0 exponential, 1 gaussian, 2 median (decoder.py:260-331)
    int window;                  // ring slots of methods 1 and 2
    const float* sm_param;       // alpha (method 0) or gaussian_weights (window) (method 1)
    float* state;                // method 0: (B, NB); 1, 2: (B, window * NB + 1) = ring (window, NB) + slot pointer
    int constraints;
    float* out; float* raw;
};

// Thread q of the workgroup owns blendshape q of batch element b; z = its decoder output before the activation.  Threads
// q >= 64 only take part in the barrier.  ys: 64 floats of LDS.
__device__ __forceinline__ void kmm_tail_dev(const KmmTail& a, int64_t b, int q, float z, float* ys) {
    const int NB = a.NB;
    float y = 0.f;
    if (q < NB) {
        y = a.out_act == 0 ? 1.0f / (1.0f + expf(-z)) : (a.out_act == 1 ? tanhf(z) : z);
        if (a.prev) y = (1.0f - 0.1f) * y + 0.1f * a.prev[b * NB + q];      // decoder.py:172-175
        if (a.raw) a.raw[b * NB + q] = y;
        if (a.smooth == 0) {                                                   // decoder.py:278-292
            const float alpha = 1.0f / (1.0f + expf(-a.sm_param[0]));
            y = alpha * a.state[b * NB + q] + (1.0f - alpha) * y;
            a.state[b * NB + q] = y;
        } else if (a.smooth > 0) {                                             // decoder.py:294-340: history ring, one slot per call
            float* ring = a.state + b * ((int64_t)a.window * NB + 1);
            // the slot pointer lives in the CALLER's float state: anything but 0 .. window - 1 (a state that was not zeroed,
            // a NaN) restarts at slot 0 instead of indexing the ring with it
            const float pf = ring[(int64_t)a.window * NB];                      // every thread reads it before thread 0 moves it
            const int ptr = (pf >= 0.f && pf < (float)a.window) ? (int)pf : 0;
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = k < a.window ? (k == ptr ? y : ring[k * NB + q]) : 0.f;
            ring[ptr * NB + q] = y;
            if (a.smooth == 1) {          // softmax of the learnable weights over the SLOTS (:307-317)
                float m = -INFINITY, s = 0.f, acc = 0.f;
                for (int k = 0; k < a.window; ++k) m = fmaxf(m, a.sm_param[k]);
                for (int k = 0; k < a.window; ++k) s += expf(a.sm_param[k] - m);
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < a.window) acc += (expf(a.sm_param[k] - m) / s) * v[k];
                y = acc;
            } else {                      // torch.median(dim=0): the lower middle of the sorted slots; NaN if any slot is NaN
                bool nan = false;
#pragma unroll
                for (int k = 0; k < 16; ++k) nan = nan || (k < a.window && v[k] != v[k]);
#pragma unroll
                for (int i = 1; i < 16; ++i)          // insertion sort of the first `window` entries (fully unrolled: registers)
#pragma unroll
                    for (int j = i; j > 0; --j)
                        if (i < a.window && v[j] < v[j - 1]) { const float t = v[j]; v[j] = v[j - 1]; v[j - 1] = t; }
                float med = v[0];
#pragma unroll
                for (int k = 0; k < 16; ++k) med = k == (a.window - 1) / 2 ? v[k] : med;
                y = nan ? NAN : med;
            }
        }
        if (a.constraints) y = y < 0.f ? 0.f : (y > 1.f ? 1.f : y);          // decoder.py:434-438; NaN stays NaN, as torch.clamp
    }
    if (q < 64) ys[q] = y;
    __syncthreads();
    if (a.smooth > 0 && q == 0) {       // after every thread of the element has read the pointer
        float* pp = a.state + b * ((int64_t)a.window * NB + 1) + (int64_t)a.window * NB;
        const float pf = pp[0];
        const int ptr = (pf >= 0.f && pf < (float)a.window) ? (int)pf : 0;
        pp[0] = (float)((ptr + 1) % a.window);
    }
    if (q < NB) {
        if (a.constraints) {
            const int pa[2] = {25, 20}, pb[2] = {26, 21};                     // decoder.py:384-387, :451-456
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (q == pa[i] || q == pb[i]) y = ys[q] / ((ys[pa[i]] + ys[pb[i]]) + 1e-8f);
        }
        a.out[b * NB + q] = y;
    }
}
