// mixer.hip -- the mixer of SURVEY 8(f) on the device: mix_waveforms (src/mixer.cpp:133-140) over the batches the
// demod entry points produced, in the jitter-free input order of mixer_thread (src/mixer.cpp:190-213).
// Streaming and HBM-bound: per output sample it reads one float per signalling input and writes one (two) floats;
// one thread per 4 consecutive samples (16 B per lane per row), inputs walked in order so the float sum is the
// reference's left-to-right `sum[s] += in[s] * mult`.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi_airband.h"
#include "plan.hpp"

namespace mi {
std::string& last_error_ref();  // mi_airband.cpp
}

struct MixIn {
    int row;
    float ml, mr;  // ampfactor * ampl, ampfactor * ampr (mixer.cpp:203,206)
};

struct mi_mixer {
    int gpu = 0;
    int n = 0;
    int stereo = 0;
    MixIn* d_in = nullptr;
};

namespace {

__global__ __launch_bounds__(256) void k_mix(const MixIn* __restrict__ in, const int n, const int stereo, const float* __restrict__ wave,
                                             const size_t row_stride, const char* __restrict__ axc, const size_t axc_stride, const int nbatches,
                                             float* __restrict__ left, float* __restrict__ right, char* __restrict__ axc_out) {
    const size_t q = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;  // group of 4 samples
    const size_t total = static_cast<size_t>(nbatches) * (mi::kWaveBatch / 4);
    if (q >= total)
        return;
    const int batch = static_cast<int>(q / (mi::kWaveBatch / 4));  // 2000 % 4 == 0: a group lies in one batch
    float4 l = make_float4(0.f, 0.f, 0.f, 0.f), r = l;             // memset(channel->waveout, 0, ...), mixer.cpp:194-197
    bool any = false;
    for (int j = 0; j < n; ++j) {
        const MixIn m = in[j];
        if (axc[static_cast<size_t>(m.row) * axc_stride + batch] == MI_NO_SIGNAL)  // input->has_signal, output.cpp:564
            continue;
        any = true;
        const float4 v = *reinterpret_cast<const float4*>(wave + static_cast<size_t>(m.row) * row_stride + 4 * q);
        if (m.ml != 0.0f) {  // mix_waveforms returns at once for mult == 0.0f
            l.x += v.x * m.ml, l.y += v.y * m.ml, l.z += v.z * m.ml, l.w += v.w * m.ml;
        }
        if (stereo && m.mr != 0.0f) {
            r.x += v.x * m.mr, r.y += v.y * m.mr, r.z += v.z * m.mr, r.w += v.w * m.mr;
        }
    }
    *reinterpret_cast<float4*>(left + 4 * q) = l;
    if (stereo)
        *reinterpret_cast<float4*>(right + 4 * q) = r;
    if (q % (mi::kWaveBatch / 4) == 0)
        axc_out[batch] = any ? MI_SIGNAL : MI_NO_SIGNAL;
}

int mfail(int code, const std::string& msg) {
    mi::last_error_ref() = msg;
    return code;
}

}  // namespace

extern "C" {

int mi_mixer_create(const mi_mix_input* inputs, int ninputs, int gpu, mi_mixer** out) {
    if (!inputs || !out || ninputs < 1)
        return mfail(MI_ERR_INVALID, "mixer needs at least one input");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return mfail(MI_ERR_NO_DEVICE, "no HIP device: the mixer runs on the GPU only");
    if (gpu < 0 || gpu >= ndev)
        return mfail(MI_ERR_INVALID, "gpu index out of range");
    std::vector<MixIn> v(static_cast<size_t>(ninputs));
    int stereo = 0;
    for (int i = 0; i < ninputs; ++i) {
        const mi_mix_input& k = inputs[i];
        if (k.row < 0)
            return mfail(MI_ERR_INVALID, "mixer input row must be >= 0");
        if (!(k.balance >= -1.0f && k.balance <= 1.0f))
            return mfail(MI_ERR_INVALID, "balance out of allowed range <-1.0;1.0>");  // config.cpp:183-186
        const float ampl = fminf(1.0f, 1.0f - k.balance), ampr = fminf(1.0f, 1.0f + k.balance);  // mixer.cpp:80-81
        v[i].row = k.row;
        v[i].ml = k.ampfactor * ampl;
        v[i].mr = k.ampfactor * ampr;
        if (k.balance != 0.0f)
            stereo = 1;  // mixer.cpp:82-83
    }
    mi_mixer* m = new mi_mixer();
    m->gpu = gpu;
    m->n = ninputs;
    m->stereo = stereo;
    if (hipSetDevice(gpu) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&m->d_in), v.size() * sizeof(MixIn)) != hipSuccess ||
        hipMemcpy(m->d_in, v.data(), v.size() * sizeof(MixIn), hipMemcpyHostToDevice) != hipSuccess) {
        if (m->d_in)
            (void)hipFree(m->d_in);
        delete m;
        return mfail(MI_ERR_HIP, "mixer: device allocation failed");
    }
    *out = m;
    return MI_OK;
}

void mi_mixer_destroy(mi_mixer* m) {
    if (!m)
        return;
    (void)hipSetDevice(m->gpu);
    if (m->d_in)
        (void)hipFree(m->d_in);
    delete m;
}

int mi_mixer_is_stereo(const mi_mixer* m) {
    return m ? m->stereo : 0;
}

int mi_mixer_process_device(mi_mixer* m, const float* d_waveout, size_t row_stride, const char* d_axc, size_t axc_stride, int nbatches,
                            float* d_left, float* d_right, char* d_axc_out, void* hip_stream) {
    if (!m || !d_waveout || !d_axc || !d_left || !d_axc_out || nbatches < 1)
        return mfail(MI_ERR_INVALID, "NULL argument");
    if (m->stereo && !d_right)
        return mfail(MI_ERR_INVALID, "stereo mixer needs a right-channel buffer");
    if (row_stride % 4 != 0 || reinterpret_cast<uintptr_t>(d_waveout) % 16 != 0 || reinterpret_cast<uintptr_t>(d_left) % 16 != 0 ||
        (d_right && reinterpret_cast<uintptr_t>(d_right) % 16 != 0))
        return mfail(MI_ERR_INVALID, "audio buffers must be 16-byte aligned with a row stride that is a multiple of 4 floats");
    if (hipSetDevice(m->gpu) != hipSuccess)
        return mfail(MI_ERR_HIP, "hipSetDevice failed");
    const size_t total = static_cast<size_t>(nbatches) * (mi::kWaveBatch / 4);
    hipLaunchKernelGGL(k_mix, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), m->d_in, m->n,
                       m->stereo, d_waveout, row_stride, d_axc, axc_stride, nbatches, d_left, d_right, d_axc_out);
    if (hipGetLastError() != hipSuccess)
        return mfail(MI_ERR_HIP, "mixer kernel launch failed");
    return MI_OK;
}

}  // extern "C"
