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
int share_engines = 1;

#define SLEEP(ms) usleep((ms)*1000)

static input_t* input_new_impl(sample_format_t sfmt, int sample_rate, int centerfreq, int pinned);

input_t* input_new_for_format(sample_format_t sfmt, int sample_rate, int centerfreq) {
    return input_new_impl(sfmt, sample_rate, centerfreq, 0);
}

input_t* input_new_pinned_for_format(sample_format_t sfmt, int sample_rate, int centerfreq) {
    return input_new_impl(sfmt, sample_rate, centerfreq, 1);
}

static input_t* input_new_impl(sample_format_t sfmt, int sample_rate, int centerfreq, int pinned) {
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
    const size_t ring_bytes = in->buf_size + 2 * in->bytes_per_sample * fft_size;
    in->buffer = pinned ? (unsigned char*)mi_host_alloc(ring_bytes) : NULL;
    in->buffer_pinned = in->buffer != NULL;
    if (in->buffer)
        memset(in->buffer, 0, ring_bytes);
    else
        in->buffer = (unsigned char*)calloc(sizeof(unsigned char), ring_bytes);
    in->bufs = in->bufe = 0;
    in->overflow_count = 0;
    in->state = INPUT_INITIALIZED;
    pthread_mutex_init(&in->buffer_lock, NULL);
    return in;
}

void input_free(input_t* in) {
    if (!in)
        return;
    if (in->buffer_pinned)
        mi_host_free(in->buffer);
    else
        free(in->buffer);
    pthread_mutex_destroy(&in->buffer_lock);
    free(in);
}

namespace {

struct RingGuard {  // buffer_lock for the length of a scope
    explicit RingGuard(input_t* in) : m(&in->buffer_lock) { pthread_mutex_lock(m); }
    ~RingGuard() { pthread_mutex_unlock(m); }
    pthread_mutex_t* m;
};

size_t tail_mirror_len(const input_t* in) {  // bytes kept behind buf_size as a copy of the ring's first bytes
    return 2 * static_cast<size_t>(in->bytes_per_sample) * fft_size;
}

}  // namespace

// The RX side of the ring.  Behaviour follows input-helpers.cpp:37-63 (the spec; tests/test_host_mirror.py compares every
// byte and counter with a model of that rule): the bytes go in at the write position, in two pieces when they run past
// buf_size; whatever piece landed at offset 0 is then copied, up to the mirror's length, behind buf_size, so that a window
// starting near the end of the ring reads on without wrapping; the write position advances modulo buf_size; passing the
// read position counts one overflow.
void circbuffer_append(input_t* const input, unsigned char* buf, size_t len) {
    if (!len)
        return;
    RingGuard hold(input);
    const size_t start = input->bufe;
    const size_t until_end = input->buf_size - start;
    const size_t first = len <= until_end ? len : until_end;  // piece that fits before buf_size
    const size_t second = len - first;                        // piece that continues at offset 0
    memcpy(input->buffer + start, buf, first);
    size_t at_front = 0;  // how many freshly written bytes begin at offset 0
    if (second) {
        memcpy(input->buffer, buf + first, second);
        at_front = second;
    } else if (start == 0) {
        at_front = first;
    }
    if (at_front)
        memcpy(input->buffer + input->buf_size, input->buffer, std::min(at_front, tail_mirror_len(input)));
    input->bufe = (start + len) % input->buf_size;
    if (start < input->bufs && input->bufe >= input->bufs) {
        fprintf(stderr, "Warning: buffer overflow\n");
        input->overflow_count++;
    }
}

size_t ring_fill(input_t* in) {
    RingGuard hold(in);
    return in->bufe >= in->bufs ? in->bufe - in->bufs : in->buf_size - in->bufs + in->bufe;
}

const unsigned char* ring_contiguous(const input_t* in, size_t need, std::vector<unsigned char>& scratch) {
    // (The mirrored tail is not relied on here: the reference's rule leaves part of it stale when a short append follows a
    // wrap, and a batch is far longer than the tail anyway.)
    if (in->bufs + need <= in->buf_size)
        return in->buffer + in->bufs;
    const size_t before_wrap = in->buf_size - in->bufs;
    scratch.resize(need);
    memcpy(scratch.data(), in->buffer + in->bufs, before_wrap);
    memcpy(scratch.data() + before_wrap, in->buffer, need - before_wrap);
    return scratch.data();
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
    dev->engine_stream = 0;
    dev->engine_streams = 1;
    dev->engine_owner = 0;
    return dev;
}

std::vector<mi_channel_cfg> channel_cfgs_of(const device_t* dev) {
    std::vector<mi_channel_cfg> out;
    out.reserve((size_t)dev->channel_count);
    for (int i = 0; i < dev->channel_count; i++)
        out.push_back(dev->channels[i].cfg);
    return out;
}

void device_free(device_t* dev) {
    if (!dev)
        return;
    if (dev->engine && dev->engine_owner)
        mi_demod_destroy(dev->engine);
    for (int i = 0; i < dev->channel_count; i++)
        free(dev->channels[i].freqlist);
    free(dev->channels);
    free(dev);
}

namespace {

mi_device_cfg device_cfg_of(const device_t* dev) {
    mi_device_cfg dc;
    dc.sample_rate = dev->input->sample_rate;
    dc.centerfreq = dev->input->centerfreq;
    dc.fft_size_log = (int)fft_size_log;
    dc.sfmt = (int)dev->input->sfmt;
    dc.fullscale = dev->input->fullscale;
    dc.tau = dev->tau;
    dc.fm_quadri = fm_quadri_demod_selected;
    return dc;
}

bool same_channel(const mi_channel_cfg& a, const mi_channel_cfg& b) {
    return a.freq == b.freq && a.modulation == b.modulation && a.squelch_threshold_dbfs == b.squelch_threshold_dbfs &&
           a.has_snr_threshold == b.has_snr_threshold && a.squelch_snr_db == b.squelch_snr_db && a.notch_freq == b.notch_freq && a.notch_q == b.notch_q &&
           a.ctcss_freq == b.ctcss_freq && a.bandwidth == b.bandwidth && a.ampfactor == b.ampfactor && a.tau == b.tau && a.afc == b.afc &&
           a.has_iq_outputs == b.has_iq_outputs;
}

// Two devices may be streams of one engine iff the engine would derive the same plan for both.
bool same_plan(const device_t* a, const device_t* b) {
    const mi_device_cfg x = device_cfg_of(a), y = device_cfg_of(b);
    if (x.sample_rate != y.sample_rate || x.centerfreq != y.centerfreq || x.sfmt != y.sfmt || x.fullscale != y.fullscale || x.tau != y.tau ||
        a->channel_count != b->channel_count)
        return false;
    for (int i = 0; i < a->channel_count; i++)
        if (!same_channel(a->channels[i].cfg, b->channels[i].cfg))
            return false;
    return true;
}

}  // namespace

int init_demod(demod_params_t* params, Signal* signal, int device_start, int device_end, int gpu) {
    params->mp3_signal = signal;
    params->device_start = device_start;
    params->device_end = device_end;
    for (int d = device_start; d < device_end; d++) {
        device_t* dev = devices + d;
        if (dev->engine)
            continue;  // already a stream of an earlier device's engine
        std::vector<int> members = {d};
        for (int e = d + 1; share_engines && e < device_end; e++)
            if (!devices[e].engine && same_plan(dev, devices + e))
                members.push_back(e);
        const mi_device_cfg dc = device_cfg_of(dev);
        std::vector<mi_channel_cfg> cc = channel_cfgs_of(dev);
        mi_demod* engine = NULL;
        int rc = mi_demod_create(&dc, cc.data(), dev->channel_count, (int)members.size(), 1, gpu, &engine);
        if (rc == MI_OK)
            rc = mi_demod_prepare(engine, 1);  // staging + a rehearsal now: the first batch must not pay for them
        if (rc != MI_OK) {
            fprintf(stderr, "init_demod: device %d: %s\n", d, mi_last_error());
            if (engine)
                mi_demod_destroy(engine);
            return rc;
        }
        for (size_t k = 0; k < members.size(); k++) {
            device_t* m = devices + members[k];
            m->engine = engine;
            m->engine_stream = (int)k;
            m->engine_streams = (int)members.size();
            m->engine_owner = k == 0;
        }
    }
    return 0;
}

namespace {

// The engines one demod thread serves, visited in turn (the reference walks its devices [device_start, device_end) the same
// way, rtl_airband.cpp:300-306); an engine's devices are its streams, in stream order.
struct EngineGroup {
    mi_demod* engine;
    std::vector<int> members;
};

std::vector<EngineGroup> engine_groups(int first, int end) {
    std::vector<EngineGroup> groups;
    for (int d = first; d < end; d++) {
        EngineGroup* g = NULL;
        for (EngineGroup& x : groups)
            if (x.engine == devices[d].engine)
                g = &x;
        if (!g) {
            groups.push_back(EngineGroup{devices[d].engine, {}});
            g = &groups.back();
        }
        if ((int)g->members.size() <= devices[d].engine_stream)
            g->members.resize((size_t)devices[d].engine_stream + 1, -1);
        g->members[(size_t)devices[d].engine_stream] = d;
    }
    return groups;
}

// Scratch of one demod thread: what one engine call returns before it is published into channel_t.
struct BatchScratch {
    std::vector<std::vector<unsigned char>> linear;  // per stream: a batch that wraps in its ring, assembled
    std::vector<unsigned char> silence;              // what a retired device's stream is fed
    std::vector<float> wave, iq;
    std::vector<char> axc;
    std::vector<mi_channel_stats> stats;
    void fit(int nstreams, int nch) {
        const size_t rows = (size_t)nstreams * (size_t)nch;
        linear.resize((size_t)nstreams);
        wave.resize(rows * (WAVE_BATCH + AGC_EXTRA));
        iq.resize(rows * WAVE_BATCH * 2);
        axc.resize(rows);
        stats.resize(rows);
    }
};

// channel_t / freq_t after a batch, exactly what the reference's loop leaves: waveout[0, WAVE_BATCH) final and
// [WAVE_BATCH, +AGC_EXTRA) lookahead, iq_out, axcindicate, the squelch statistics and counters (rtl_airband.cpp:612-669).
// `stream`: which stream of the engine's output arrays this device is.
void publish_batch(device_t* dev, const BatchScratch& b, int stream) {
    for (int i = 0; i < dev->channel_count; i++) {
        channel_t* ch = dev->channels + i;
        freq_t* f = ch->freqlist + ch->freq_idx;
        const size_t k = (size_t)stream * (size_t)dev->channel_count + (size_t)i;
        memcpy(ch->waveout, b.wave.data() + k * (WAVE_BATCH + AGC_EXTRA), (WAVE_BATCH + AGC_EXTRA) * sizeof(float));
        if (ch->has_iq_outputs)
            memcpy(ch->iq_out, b.iq.data() + k * WAVE_BATCH * 2, WAVE_BATCH * 2 * sizeof(float));
        ch->axcindicate = (status)b.axc[k];
        f->squelch = b.stats[k];
        f->agcavgfast = b.stats[k].agcavgfast;
        f->active_counter = (size_t)b.stats[k].active_counter;
    }
    __sync_synchronize();  // the output thread reads these without a lock once it sees waveavail (output.cpp:933-950)
    if (dev->waveavail == 1)
        dev->output_overrun_count++;  // the previous batch was not collected (rtl_airband.cpp:671-676)
    else
        dev->waveavail = 1;
}

}  // namespace

// The demod thread.  Control flow of the reference's loop (rtl_airband.cpp:381-422, 671-691) with one engine call per
// WAVE_BATCH in place of the per-window body: exit flag, "all receivers failed", skipping inputs that are not running
// (a failed one is retired once), the availability rule, the 10 ms nap when the ring is short, publish, signal, advance.
// The unit of a turn is an engine: its devices (equal plans, hence equal sample rates: their rings fill in step) are served by
// ONE submit / wait pair, each device's batch uploaded from its own ring; a turn happens when every running device of the
// engine has a batch.  A retired device's stream is fed silence (streams are independent: nobody else notices).
void* demodulate(void* params) {
    demod_params_t* const dp = (demod_params_t*)params;
    std::vector<EngineGroup> groups = engine_groups(dp->device_start, dp->device_end);
    BatchScratch scratch;
    size_t at = 0;
    auto next = [&]() { at = (at + 1 < groups.size()) ? at + 1 : 0; };
    for (; !do_exit && !groups.empty();) {
        EngineGroup& g = groups[at];
        if (devices_running == 0) {
            fprintf(stderr, "All receivers failed, exiting\n");
            do_exit = 1;
            break;
        }
        const int nstreams = (int)g.members.size();
        const size_t consumed = mi_demod_bytes_consumed(g.engine, 1);
        const size_t needed = mi_demod_bytes_needed(g.engine, 1);
        int running = 0;
        bool short_ring = false;
        for (int m : g.members) {
            input_t* const in = devices[m].input;
            if (in->state != INPUT_RUNNING) {
                if (in->state == INPUT_FAILED) {  // retire it: its outputs stay silent from now on
                    in->state = INPUT_DISABLED;
                    devices_running--;
                }
                continue;
            }
            running++;
            // The reference starts a window when the ring holds one hop plus one window (rtl_airband.cpp:417).  Applied to the
            // LAST window of a batch: everything the batch consumes, plus one window.
            if (ring_fill(in) < consumed + fft_size * (size_t)in->bytes_per_sample * 2)
                short_ring = true;
        }
        if (running == 0) {
            next();
            continue;
        }
        if (short_ring) {
            next();
            SLEEP(10);
            continue;
        }
        const int nch = devices[g.members[0]].channel_count;
        scratch.fit(nstreams, nch);
        std::vector<const uint8_t*> streams((size_t)nstreams);
        for (int k = 0; k < nstreams; k++) {
            input_t* const in = devices[g.members[(size_t)k]].input;
            if (in->state == INPUT_RUNNING) {
                streams[(size_t)k] = ring_contiguous(in, needed, scratch.linear[(size_t)k]);
            } else {
                scratch.silence.assign(needed, in->sfmt == SFMT_U8 ? 0x80 : 0);
                streams[(size_t)k] = scratch.silence.data();
            }
        }
        int rc = mi_demod_submit(g.engine, streams.data(), 1, scratch.wave.data(), scratch.iq.data(), scratch.axc.data(), scratch.stats.data());
        if (rc == MI_OK)
            rc = mi_demod_wait(g.engine);
        if (rc != MI_OK) {  // an engine failure at run time is an input failure of its devices
            fprintf(stderr, "demodulate: engine of device %d: %s\n", g.members[0], mi_last_error());
            for (int m : g.members)
                if (devices[m].input->state == INPUT_RUNNING)
                    devices[m].input->state = INPUT_FAILED;
            continue;
        }
        for (int k = 0; k < nstreams; k++) {
            device_t* const dev = devices + g.members[(size_t)k];
            input_t* const in = dev->input;
            if (in->state != INPUT_RUNNING)
                continue;
            publish_batch(dev, scratch, k);
            in->bufs = (in->bufs + consumed) % in->buf_size;  // rtl_airband.cpp:691
        }
        dp->mp3_signal->send();  // rtl_airband.cpp:684
        next();
    }
    return NULL;
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
