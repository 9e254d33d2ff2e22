// GPU side of the training data feed (SURVEY.md section 8f "next" #4): one kernel per minibatch replaces the
// per-item CPU chain of rave/dataset.py:
//   int16 PCM -> float32 / (2^15 - 1)                      (AudioDataset.__getitem__, dataset.py:75-78)
//   RandomCrop(n_signal)                                   (transforms.py:96-106; crop points drawn by the host)
//   RandomApply(random_phase_mangle, p = .8)               (dataset.py:223-226,283-299: all-pass biquad with a random
//                                                           pole angle, scipy.signal.lfilter = direct form II transposed,
//                                                           evaluated in float64 like scipy does for float64 taps)
//   Dequantize(16): x += U[0,1) / 2^16, then float32       (transforms.py:109-115, dataset.py:246)
// At > 1e8 samples/s per GPU the 8-worker scipy chain cannot keep up; here a (clip, channel) row is one workgroup:
// 4096-sample chunks are staged in LDS, thread 0 advances the two-state recurrence, all threads add the noise and
// store.  The recurrence is sequential by nature (0.5 ms for 65536 samples) but rows run in parallel and the
// kernel is meant for a side stream, under the training step.
#include "common.hpp"

namespace {

constexpr int kChunk = 4096;

__global__ __launch_bounds__(256) void feed_kernel(const int16_t* __restrict__ pcm, const int64_t* __restrict__ src_offset,
                                                   const double* __restrict__ coef, const float* __restrict__ noise,
                                                   int n_signal, double quant, float* __restrict__ out) {
    __shared__ double buf[kChunk];
    const int r = blockIdx.x;
    const int16_t* src = pcm + src_offset[r];
    const double b0 = coef[r * 5 + 0], b1 = coef[r * 5 + 1], b2 = coef[r * 5 + 2], a1 = coef[r * 5 + 3], a2 = coef[r * 5 + 4];
    const bool filt = b0 == b0;                 // NaN marks "transform not applied" (RandomApply miss)
    double z0 = 0.0, z1 = 0.0;
    for (int c0 = 0; c0 < n_signal; c0 += kChunk) {
        const int len = min(kChunk, n_signal - c0);
        for (int i = threadIdx.x; i < len; i += 256)
            buf[i] = (double)((float)src[c0 + i] / 32767.0f);          // float32 division, as the reference
        __syncthreads();
        if (filt && threadIdx.x == 0) {
            for (int i = 0; i < len; ++i) {
                const double x = buf[i];
                const double y = b0 * x + z0;
                z0 = b1 * x - a1 * y + z1;
                z1 = b2 * x - a2 * y;
                buf[i] = y;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 256) {
            const long o = (long)r * n_signal + c0 + i;
            out[o] = (float)(buf[i] + (double)noise[o] * quant);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int rh_feed_batch_i16_f32(const int16_t* pcm, const int64_t* src_offset, const double* coef, const float* noise,
                                     int32_t rows, int32_t n_signal, int32_t bit_depth, float* out, rh_stream_t stream) {
    RH_REQUIRE(rows >= 0 && n_signal > 0 && bit_depth > 0 && bit_depth < 32, RH_ERR_INVALID, "feed_batch: bad sizes");
    if (rows == 0) return RH_OK;
    RH_REQUIRE(pcm && src_offset && coef && noise && out, RH_ERR_INVALID, "feed_batch: null pointer");
    hipLaunchKernelGGL(feed_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pcm, src_offset, coef, noise, n_signal,
                       1.0 / (double)(1ll << bit_depth), out);
    return rh_check_launch("feed_batch");
}
