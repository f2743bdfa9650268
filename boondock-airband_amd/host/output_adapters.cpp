#include "output_adapters.hpp"

#include <cstring>

int rawfile_put(rawfile_out_t* out, const float* iq_out, char axcindicate) {
    if (!out->continuous && axcindicate == NO_SIGNAL && !out->active)
        return 0;  // output.cpp:516-519 (the file would be closed here if a transmission just ended)
    const size_t buflen = 2 * sizeof(float) * WAVE_BATCH;  // output.cpp:548-551
    const size_t written = fwrite(iq_out, 1, buflen, out->f);
    out->active = (axcindicate != NO_SIGNAL);  // output.cpp:560
    if (written < buflen)
        return -1;
    out->batches_written++;
    return 1;
}

bool udp_stream_sends(bool continuous, char axcindicate) {
    return continuous || axcindicate != NO_SIGNAL;
}

size_t udp_payload_bytes(bool stereo) {
    return (size_t)WAVE_BATCH * sizeof(float) * (stereo ? 2 : 1);
}

size_t udp_payload_mono(const float* waveout, unsigned char* out) {
    memcpy(out, waveout, (size_t)WAVE_BATCH * sizeof(float));
    return (size_t)WAVE_BATCH * sizeof(float);
}

size_t udp_payload_stereo(const float* waveout, const float* waveout_r, unsigned char* out) {
    // The reference interleaves into stereo_buffer and sends `len * 2` bytes with len = WAVE_BATCH * sizeof(float):
    // 2 * WAVE_BATCH floats = WAVE_BATCH frames of (left, right).
    float* o = reinterpret_cast<float*>(out);
    for (size_t i = 0; i < (size_t)WAVE_BATCH; ++i) {
        o[2 * i] = waveout[i];
        o[2 * i + 1] = waveout_r[i];
    }
    return (size_t)WAVE_BATCH * sizeof(float) * 2;
}
