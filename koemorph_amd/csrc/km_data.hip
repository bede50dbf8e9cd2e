// km_data.hip -- device-side window producer for sequence training.
//
// The reference's KoeMorphSequentialDataset (src/data/sequential_dataset.py:136-209) keeps every clip on the host and,
// per window, slices 136 448 samples + 256 label rows, converts them to tensors and ships them to the GPU; labels
// recorded at 60 fps are first resampled to 30 fps with np.linspace + np.interp (:136-154).  Here the clip and its
// labels live in HBM; windows are gathered by start frame on the device and the resampling runs there too, in the
// same float64 arithmetic numpy uses (bit-identical results, tested against numpy itself).
#include <hip/hip_runtime.h>

#include "km_context.h"

namespace km {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// dst[t, k] = float32(np.interp(np.linspace(0, n_src - 1, n_dst)[t], np.arange(n_src), src[:, k]))
//   np.linspace: step = (n_src - 1) / (n_dst - 1) in float64, x_t = t * step, last point = n_src - 1 exactly
//   np.interp (numpy/_core/src/multiarray/compiled_base.c): x beyond the last knot -> fp[-1]; x == xp[j] -> fp[j];
//   otherwise slope * (x - xp[j]) + fp[j] with slope = (fp[j+1] - fp[j]) / (xp[j+1] - xp[j]), separate multiply and
//   add in float64 (no fused multiply-add), then the float64 -> float32 store of `resampled[:, i] = ...`
__global__ void resample_labels_kernel(const float* __restrict__ src, int64_t n_src, int dims, int64_t n_dst,
                                       float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dst * dims) return;
    const int64_t t = i / dims;
    const int k = (int)(i - t * dims);
    double x;
    if (n_dst == 1) x = 0.0;                                        // linspace(0, stop, 1) = [0.]
    else if (t == n_dst - 1) x = (double)(n_src - 1);
    else x = __dmul_rn((double)t, __ddiv_rn((double)(n_src - 1), (double)(n_dst - 1)));
    const int64_t j = (int64_t)x;                                   // xp = arange: the knot index is floor(x), x >= 0
    double r;
    if (j >= n_src - 1) r = (double)src[(n_src - 1) * dims + k];
    else {
        const double f0 = (double)src[j * dims + k], f1 = (double)src[(j + 1) * dims + k];
        if ((double)j == x) r = f0;
        else r = __dadd_rn(__dmul_rn(__dsub_rn(f1, f0), __dsub_rn(x, (double)j)), f0);
    }
    dst[i] = (float)r;
}

// audio_out[b, :] = clip[start_frame[b] * hop : + window_samples]   (sequential_dataset.py:181-187)
// labels_out[b, :, :] = labels[start_frame[b] : + window_frames]     (:188)
// target_out[b, :] = labels[start_frame[b] + window_frames - 1]      (the frame the window's prediction belongs to)
__global__ void gather_windows_kernel(const float* __restrict__ clip, int64_t clip_len, const int32_t* __restrict__ start_frames,
                                      int hop, int64_t window_samples, float* __restrict__ audio_out,
                                      const float* __restrict__ labels, int64_t n_label_frames, int window_frames, int dims,
                                      float* __restrict__ labels_out, float* __restrict__ target_out) {
    const int b = blockIdx.y;
    const int64_t s0 = (int64_t)start_frames[b] * hop;
    if (audio_out) {
        float* o = audio_out + (int64_t)b * window_samples;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < window_samples; i += (int64_t)gridDim.x * blockDim.x)
            o[i] = s0 + i < clip_len ? clip[s0 + i] : 0.f;
    }
    if (labels) {
        const int64_t f0 = start_frames[b];
        const int64_t nl = (int64_t)window_frames * dims;
        if (labels_out) {
            float* o = labels_out + (int64_t)b * nl;
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (int64_t)gridDim.x * blockDim.x) {
                const int64_t fr = f0 + i / dims;
                o[i] = fr < n_label_frames ? labels[fr * dims + i % dims] : 0.f;
            }
        }
        if (target_out && blockIdx.x == 0) {
            const int64_t fr = f0 + window_frames - 1;
            for (int i = threadIdx.x; i < dims; i += blockDim.x)
                target_out[(int64_t)b * dims + i] = fr < n_label_frames ? labels[fr * dims + i] : 0.f;
        }
    }
}

}  // namespace km

using namespace km;

extern "C" {

int km_resample_labels(const float* src_dev, int64_t n_src, int32_t dims, int64_t n_dst, float* dst_dev, void* stream) {
    if (!src_dev || !dst_dev || n_src <= 0 || dims <= 0 || n_dst < 0)
        return fail(KM_ERR_INVALID_ARG, "km_resample_labels: bad argument");
    if (n_dst == 0) return KM_OK;
    const int64_t n = n_dst * dims;
    hipLaunchKernelGGL(resample_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_dev, n_src,
                       dims, n_dst, dst_dev);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

int km_gather_windows(const float* clip_dev, int64_t clip_len, const int32_t* start_frames_dev, int64_t B, int32_t hop,
                      int64_t window_samples, float* audio_out_dev, const float* labels_dev, int64_t n_label_frames,
                      int32_t window_frames, int32_t dims, float* labels_out_dev, float* target_out_dev, void* stream) {
    if (!start_frames_dev || B <= 0 || hop <= 0 || (audio_out_dev && (!clip_dev || window_samples <= 0 || clip_len < 0)) ||
        ((labels_out_dev || target_out_dev) && (!labels_dev || window_frames <= 0 || dims <= 0)))
        return fail(KM_ERR_INVALID_ARG, "km_gather_windows: bad argument");
    if (B > 65535) return fail(KM_ERR_INVALID_ARG, "km_gather_windows: at most 65535 windows per call");
    hipLaunchKernelGGL(gather_windows_kernel, dim3(64, (unsigned)B), dim3(256), 0, (hipStream_t)stream, clip_dev, clip_len,
                       start_frames_dev, hop, window_samples, audio_out_dev, labels_dev, n_label_frames, window_frames, dims,
                       labels_out_dev, target_out_dev);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

}  // extern "C"
