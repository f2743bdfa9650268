// airband_host.hpp -- host-side mirror of the reference's interface for the demodulate() path.
//
// Same names, field meanings and error behaviour as the reference so that the existing RX threads
// upstream and output/mixer threads downstream keep working on these structs unchanged:
//   input_t            src/input-common.h:39-57      (byte ring: buffer, bufs, bufe, buf_size, buffer_lock, state, sfmt ...)
//   circbuffer_append  src/input-helpers.cpp:37-63   (wrap handling + tail pad copy)
//   freq_t/channel_t/device_t/demod_params_t/Signal  src/boondock_airband.h:210-326 (hot-path subset)
//   demodulate()       src/rtl_airband.cpp:308-694   (loop skeleton: round-robin, state checks, ring accounting,
//                                                      waveavail handoff) -- the DSP itself is one mi_demod_process() call
//   output_consume()   src/output.cpp:933-950        (what the output thread does to the contract; used by the replay tool/tests)
// Only the members the hot path touches are present; everything about outputs, mixers, labels, lame ... is not.
#pragma once
#include <pthread.h>

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

#include "../../include/mi_airband.h"

#define WAVE_RATE MI_WAVE_RATE
#define WAVE_BATCH MI_WAVE_BATCH
#define AGC_EXTRA MI_AGC_EXTRA
#define WAVE_LEN (2 * WAVE_BATCH + AGC_EXTRA)
#define MIN_BUF_SIZE 2560000

typedef enum { SFMT_UNDEF = 0, SFMT_U8, SFMT_S8, SFMT_S16, SFMT_F32 } sample_format_t;
typedef enum { INPUT_UNKNOWN = 0, INPUT_INITIALIZED, INPUT_RUNNING, INPUT_FAILED, INPUT_STOPPED, INPUT_DISABLED } input_state_t;
enum status { NO_SIGNAL = ' ', SIGNAL = '*', AFC_UP = '<', AFC_DOWN = '>' };
enum modulations { MOD_AM, MOD_NFM };

struct input_t {
    unsigned char* buffer;
    size_t buf_size, bufs, bufe;
    size_t overflow_count;
    input_state_t state;
    sample_format_t sfmt;
    float fullscale;
    int bytes_per_sample;
    int sample_rate;
    int centerfreq;
    pthread_mutex_t buffer_lock;
    int buffer_pinned;  // (mirror only) the ring lives in page-locked memory (mi_host_alloc): the engine's copy engine reads it in place
};

// The wake-up the demod thread gives the output threads once per batch (reference: class Signal, boondock_airband.h:210-230).
// Same contract -- send() wakes one waiter, wait() blocks until the next send(), a send() nobody waits for is lost --
// written over the C++11 primitives.
class Signal {
   public:
    void send() {
        std::lock_guard<std::mutex> hold(m_);
        cv_.notify_one();
    }
    void wait() {
        std::unique_lock<std::mutex> hold(m_);
        cv_.wait(hold);
    }

   private:
    std::mutex m_;
    std::condition_variable cv_;
};

struct freq_t {
    int frequency;
    float agcavgfast;
    float ampfactor;
    size_t active_counter;
    enum modulations modulation;
    mi_channel_stats squelch;  // what the stats file / TUI read through the Squelch getters, mirrored per batch
};

struct channel_t {
    float waveout[WAVE_LEN];     // [AGC_EXTRA, WAVE_BATCH+AGC_EXTRA) written per batch; the output thread reads [0, WAVE_BATCH)
    float iq_out[2 * WAVE_LEN];  // [0, 2*WAVE_BATCH) when has_iq_outputs
    status axcindicate;
    unsigned char afc;
    freq_t* freqlist;
    int freq_count, freq_idx;
    int needs_raw_iq, has_iq_outputs;
    mi_channel_cfg cfg;  // the DSP-relevant config keys (config.cpp:312-729)
};

struct device_t {
    input_t* input;
    int tau;  // device `tau` us, <0 default
    int channel_count;
    channel_t* channels;
    int waveavail;
    size_t output_overrun_count;
    mi_demod* engine;   // the MI355X engine bound to this device (init_demod); devices with equal plans share one
    int engine_stream;  // ... as its stream number `engine_stream` of `engine_streams`
    int engine_streams;
    int engine_owner;   // this device destroys the engine
};

struct demod_params_t {
    Signal* mp3_signal;
    int device_start, device_end;
};

// globals of the reference's main TU that the hot path reads (rtl_airband.cpp:71-89)
extern device_t* devices;
extern int device_count;
extern volatile int do_exit;
extern size_t fft_size_log;
extern size_t fft_size;
extern int fm_quadri_demod_selected;  // the -Q flag
extern int devices_running;

input_t* input_new_for_format(sample_format_t sfmt, int sample_rate, int centerfreq);  // ring sizing: config.cpp:799-805
// the same ring in page-locked memory (mi_host_alloc): a batch that does not wrap is uploaded straight from the ring
input_t* input_new_pinned_for_format(sample_format_t sfmt, int sample_rate, int centerfreq);
void input_free(input_t* in);
void circbuffer_append(input_t* const input, unsigned char* buf, size_t len);
// Bytes between the read and the write position of the ring (taken under buffer_lock, rtl_airband.cpp:392-397).
size_t ring_fill(input_t* in);
// `need` readable bytes starting at the read position as one contiguous run: a pointer into the ring when the run ends
// before buf_size, else the run is assembled in `scratch`.  The reference never needs this -- it consumes one window per
// turn, and a window always fits the mirrored tail behind buf_size -- a whole WAVE_BATCH per call does.
// contiguous_or_linearised().
const unsigned char* ring_contiguous(const input_t* in, size_t need, std::vector<unsigned char>& scratch);

device_t* device_new(input_t* in, const mi_channel_cfg* chans, int nch, int tau);  // channel defaults: config.cpp:271-287,319-334
// The DSP keys of a device's channels as the engine wants them (channel_cfgs_of(); in the reference tree the
// values come from the members the config.cpp patch there adds to channel_t / freq_t).
std::vector<mi_channel_cfg> channel_cfgs_of(const device_t* dev);
void device_free(device_t* dev);

// init_demod() (rtl_airband.cpp:253-266): creates the engines of the devices in [device_start, device_end).  Devices whose plans are
// equal -- same input format, rate, centre frequency, tau and channel list: the reference's "several dongles, one configuration" --
// become the streams of ONE multi-stream engine, which demodulate() serves with one submit / wait pair per turn; a device with a
// plan of its own gets an engine of its own (share_engines = 0: every device does).
// Returns 0 or a negative mi_status (the caller maps it to error() like the VideoCore codes, rtl_airband.cpp:318-332).
extern int share_engines;
int init_demod(demod_params_t* params, Signal* signal, int device_start, int device_end, int gpu);
void* demodulate(void* params);  // pthread entry, same signature as the reference

// The output thread's duty on the contract (output.cpp:933-950) for one device: if waveavail, hand each channel's
// waveout[0..WAVE_BATCH), iq_out and axcindicate to `sink`, do the AGC_EXTRA carry memcpy and clear waveavail.
typedef void (*output_sink_t)(void* user, int device, int channel, const float* waveout, const float* iq_out, char axc);
int output_consume(device_t* dev, int device_index, output_sink_t sink, void* user);
