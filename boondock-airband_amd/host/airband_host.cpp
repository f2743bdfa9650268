// airband_host.cpp -- see airband_host.hpp.  Plain host C++ (g++), links libmi_airband.so.
#include "airband_host.hpp"

#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

device_t* devices = NULL;
int device_count = 0;
volatile int do_exit = 0;
size_t fft_size_log = 9;
size_t fft_size = 1 << 9;
int fm_quadri_demod_selected = 0;
int devices_running = 0;

#define SLEEP(ms) usleep((ms)*1000)

input_t* input_new_for_format(sample_format_t sfmt, int sample_rate, int centerfreq) {
    input_t* in = (input_t*)calloc(1, sizeof(input_t));
    in->sfmt = sfmt;
    in->bytes_per_sample = (sfmt == SFMT_S16) ? 2 : (sfmt == SFMT_F32 ? 4 : 1);
    in->fullscale = (sfmt == SFMT_S16) ? 32767.5f : (sfmt == SFMT_F32 ? 1.0f : 127.5f);  // input-file.cpp:171, soapy conventions
    in->sample_rate = sample_rate;
    in->centerfreq = centerfreq;
    // config.cpp:799-805 (FFT_BATCH == 1)
    size_t fft_batch_len = 2 * in->bytes_per_sample * (size_t)ceil((double)sample_rate / (double)WAVE_RATE);
    in->buf_size = MIN_BUF_SIZE;
    if (in->buf_size % fft_batch_len != 0)
        in->buf_size += fft_batch_len - in->buf_size % fft_batch_len;
    in->buffer = (unsigned char*)calloc(sizeof(unsigned char), in->buf_size + 2 * in->bytes_per_sample * fft_size);
    in->bufs = in->bufe = 0;
    in->overflow_count = 0;
    in->state = INPUT_INITIALIZED;
    pthread_mutex_init(&in->buffer_lock, NULL);
    return in;
}

void input_free(input_t* in) {
    if (!in)
        return;
    free(in->buffer);
    pthread_mutex_destroy(&in->buffer_lock);
    free(in);
}

// input-helpers.cpp:37-63
void circbuffer_append(input_t* const input, unsigned char* buf, size_t len) {
    if (len == 0)
        return;
    pthread_mutex_lock(&input->buffer_lock);
    size_t space_left = input->buf_size - input->bufe;
    const size_t pad = 2 * input->bytes_per_sample * fft_size;
    if (space_left >= len) {
        memcpy(input->buffer + input->bufe, buf, len);
        if (input->bufe == 0)
            memcpy(input->buffer + input->buf_size, input->buffer, std::min(len, pad));
    } else {
        memcpy(input->buffer + input->bufe, buf, space_left);
        memcpy(input->buffer, buf + space_left, len - space_left);
        memcpy(input->buffer + input->buf_size, input->buffer, std::min(len - space_left, pad));
    }
    size_t old_end = input->bufe;
    input->bufe = (input->bufe + len) % input->buf_size;
    if (old_end < input->bufs && input->bufe >= input->bufs) {
        fprintf(stderr, "Warning: buffer overflow\n");
        input->overflow_count++;
    }
    pthread_mutex_unlock(&input->buffer_lock);
}

device_t* device_new(input_t* in, const mi_channel_cfg* chans, int nch, int tau) {
    device_t* dev = (device_t*)calloc(1, sizeof(device_t));
    dev->input = in;
    dev->tau = tau;
    dev->channel_count = nch;
    dev->channels = (channel_t*)calloc((size_t)nch, sizeof(channel_t));
    for (int i = 0; i < nch; i++) {
        channel_t* channel = dev->channels + i;
        for (int k = 0; k < AGC_EXTRA; k++)
            channel->waveout[k] = 0.5;  // config.cpp:321
        channel->axcindicate = NO_SIGNAL;
        channel->freq_count = 1;
        channel->freq_idx = 0;
        channel->afc = (unsigned char)chans[i].afc;
        channel->cfg = chans[i];
        channel->freqlist = (freq_t*)calloc(1, sizeof(freq_t));  // mk_freqlist, config.cpp:271-287
        channel->freqlist[0].frequency = chans[i].freq;
        channel->freqlist[0].agcavgfast = 0.5f;
        channel->freqlist[0].ampfactor = chans[i].ampfactor;
        channel->freqlist[0].modulation = chans[i].modulation == MI_MOD_NFM ? MOD_NFM : MOD_AM;
        channel->has_iq_outputs = chans[i].has_iq_outputs;
        channel->needs_raw_iq = chans[i].has_iq_outputs || chans[i].bandwidth > 0 || chans[i].modulation == MI_MOD_NFM;
    }
    dev->waveavail = 0;
    dev->output_overrun_count = 0;
    dev->engine = NULL;
    return dev;
}

void device_free(device_t* dev) {
    if (!dev)
        return;
    if (dev->engine)
        mi_demod_destroy(dev->engine);
    for (int i = 0; i < dev->channel_count; i++)
        free(dev->channels[i].freqlist);
    free(dev->channels);
    free(dev);
}

int init_demod(demod_params_t* params, Signal* signal, int device_start, int device_end, int gpu) {
    params->mp3_signal = signal;
    params->device_start = device_start;
    params->device_end = device_end;
    for (int d = device_start; d < device_end; d++) {
        device_t* dev = devices + d;
        mi_device_cfg dc;
        dc.sample_rate = dev->input->sample_rate;
        dc.centerfreq = dev->input->centerfreq;
        dc.fft_size_log = (int)fft_size_log;
        dc.sfmt = (int)dev->input->sfmt;
        dc.fullscale = dev->input->fullscale;
        dc.tau = dev->tau;
        dc.fm_quadri = fm_quadri_demod_selected;
        std::vector<mi_channel_cfg> cc((size_t)dev->channel_count);
        for (int i = 0; i < dev->channel_count; i++)
            cc[(size_t)i] = dev->channels[i].cfg;
        int rc = mi_demod_create(&dc, cc.data(), dev->channel_count, 1, 1, gpu, &dev->engine);
        if (rc != MI_OK) {
            fprintf(stderr, "init_demod: device %d: %s\n", d, mi_last_error());
            return rc;
        }
    }
    return 0;
}

static int next_device(demod_params_t* params, int current) {  // rtl_airband.cpp:300-306
    current++;
    if (current < params->device_end)
        return current;
    return params->device_start;
}

void* demodulate(void* params) {
    demod_params_t* demod_params = (demod_params_t*)params;
    std::vector<unsigned char> linear;  // a batch that wraps the ring is linearised here
    std::vector<float> wout, iqout;
    std::vector<char> axc;
    std::vector<mi_channel_stats> stats;
    size_t available;
    int device_num = demod_params->device_start;
    while (true) {
        if (do_exit)
            return NULL;
        device_t* dev = devices + device_num;
        input_t* in = dev->input;

        pthread_mutex_lock(&in->buffer_lock);
        if (in->bufe >= in->bufs)
            available = in->bufe - in->bufs;
        else
            available = in->buf_size - in->bufs + in->bufe;
        pthread_mutex_unlock(&in->buffer_lock);

        if (devices_running == 0) {  // rtl_airband.cpp:399-403
            fprintf(stderr, "All receivers failed, exiting\n");
            do_exit = 1;
            continue;
        }
        if (in->state != INPUT_RUNNING) {  // rtl_airband.cpp:405-413
            if (in->state == INPUT_FAILED) {
                in->state = INPUT_DISABLED;
                devices_running--;
            }
            device_num = next_device(demod_params, device_num);
            continue;
        }

        // The reference runs one window per loop turn and a batch completes after WAVE_BATCH of them; here a whole
        // batch is one engine call.  The availability rule of rtl_airband.cpp:417 applied to the batch's LAST window:
        // consumed bytes + one window (= what the batch reads) + one hop.
        const size_t consumed = mi_demod_bytes_consumed(dev->engine, 1);
        const size_t needed = mi_demod_bytes_needed(dev->engine, 1);
        if (available < consumed + fft_size * in->bytes_per_sample * 2) {
            device_num = next_device(demod_params, device_num);
            SLEEP(10);
            continue;
        }
        const unsigned char* src = in->buffer + in->bufs;
        if (in->bufs + needed > in->buf_size + 2 * in->bytes_per_sample * fft_size) {  // beyond the tail pad: wraps
            linear.resize(needed);
            const size_t first = in->buf_size - in->bufs;
            memcpy(linear.data(), in->buffer + in->bufs, first);
            memcpy(linear.data() + first, in->buffer, needed - first);
            src = linear.data();
        }
        const int nch = dev->channel_count;
        wout.resize((size_t)nch * (WAVE_BATCH + AGC_EXTRA));
        iqout.resize((size_t)nch * WAVE_BATCH * 2);
        axc.resize((size_t)nch);
        stats.resize((size_t)nch);
        const uint8_t* streams[1] = {src};
        int rc = mi_demod_process(dev->engine, streams, 1, wout.data(), iqout.data(), axc.data(), stats.data());
        if (rc != MI_OK) {  // a runtime engine failure is an input failure for that device
            fprintf(stderr, "demodulate: device %d: %s\n", device_num, mi_last_error());
            in->state = INPUT_FAILED;
            continue;
        }
        for (int i = 0; i < nch; i++) {
            channel_t* channel = dev->channels + i;
            freq_t* fparms = channel->freqlist + channel->freq_idx;
            // channel_t.waveout as the reference's loop leaves it: [0, WAVE_BATCH) final, [WAVE_BATCH, +AGC_EXTRA) lookahead.
            // Written before waveavail is published (the output thread reads it without a lock, output.cpp:933-950).
            memcpy(channel->waveout, wout.data() + (size_t)i * (WAVE_BATCH + AGC_EXTRA), (WAVE_BATCH + AGC_EXTRA) * sizeof(float));
            if (channel->has_iq_outputs)
                memcpy(channel->iq_out, iqout.data() + (size_t)i * WAVE_BATCH * 2, WAVE_BATCH * 2 * sizeof(float));
            channel->axcindicate = (status)axc[(size_t)i];
            fparms->squelch = stats[(size_t)i];
            fparms->agcavgfast = stats[(size_t)i].agcavgfast;
            fparms->active_counter = (size_t)stats[(size_t)i].active_counter;  // rtl_airband.cpp:667-669
        }
        __sync_synchronize();
        if (dev->waveavail == 1) {  // rtl_airband.cpp:671-676
            dev->output_overrun_count++;
        } else {
            dev->waveavail = 1;
        }
        demod_params->mp3_signal->send();  // rtl_airband.cpp:684
        in->bufs = (in->bufs + consumed) % in->buf_size;  // rtl_airband.cpp:691
        device_num = next_device(demod_params, device_num);
    }
}

int output_consume(device_t* dev, int device_index, output_sink_t sink, void* user) {
    if (!(dev->input->state == INPUT_RUNNING && dev->waveavail))  // output.cpp:933
        return 0;
    for (int j = 0; j < dev->channel_count; j++) {
        channel_t* channel = dev->channels + j;
        sink(user, device_index, j, channel->waveout, channel->has_iq_outputs ? channel->iq_out : NULL, (char)channel->axcindicate);
        memcpy(channel->waveout, channel->waveout + WAVE_BATCH, AGC_EXTRA * 4);  // output.cpp:948
    }
    __sync_synchronize();
    dev->waveavail = 0;
    return 1;
}
