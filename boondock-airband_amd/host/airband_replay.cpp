// airband_replay -- replays a raw IQ capture through the host mirror: file feed -> input_t ring
// (circbuffer_append, as input-file.cpp:82-147 does) -> demodulate() thread (engine on the GPU) ->
// output thread role (output.cpp:933-950) -> one raw f32 file per channel + the axcindicate flags.
//
// Deterministic drive: the feeder hands over less than one batch at a time and waits until the demod
// thread has starved and the output side has drained, so no batch is ever overrun and EOF drops nothing.
//
// Besides the raw audio it writes what the reference's outputs would carry for the same capture (output_adapters.hpp):
// <prefix>_ch<i>.cf32 = the O_RAWFILE byte stream of a non-continuous rawfile output (channels with has_iq_outputs),
// <prefix>_ch<i>.udp  = the concatenated datagram payloads of a non-continuous mono udp_stream output.
//
// usage: airband_replay <config.txt> <capture.iq> <out_prefix> [gpu]
//   config.txt: line 1 "sample_rate centerfreq fft_size_log sfmt tau fm_quadri"
//               then per channel "freq modulation squelch_threshold_dbfs has_snr snr_db notch notch_q ctcss bandwidth ampfactor tau afc has_iq"
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "airband_host.hpp"
#include "output_adapters.hpp"

struct Sink {
    std::vector<FILE*> audio;
    std::vector<rawfile_out_t> raw;  // f == nullptr for channels without a rawfile output
    std::vector<FILE*> udp;
    std::vector<std::string> flags;
    size_t batches = 0;
};

static void sink_fn(void* user, int /*device*/, int channel, const float* waveout, const float* iq_out, char axc) {
    Sink* s = (Sink*)user;
    const size_t c = (size_t)channel;
    fwrite(waveout, sizeof(float), WAVE_BATCH, s->audio[c]);
    s->flags[c].push_back(axc);
    if (s->raw[c].f && iq_out)
        rawfile_put(&s->raw[c], iq_out, axc);
    if (udp_stream_sends(false, axc)) {
        static thread_local std::vector<unsigned char> payload(udp_payload_bytes(false));
        const size_t n = udp_payload_mono(waveout, payload.data());
        fwrite(payload.data(), 1, n, s->udp[c]);
    }
}

// --ring-selftest <buf_size> <fft_size> <bytes_per_sample> <op>...   (no GPU needed)
//   a<len>  append len bytes (a running counter pattern)      c<n>  the consumer advances n bytes
//   r<need> print the `need` bytes ring_contiguous() hands to the engine, as hex
// After every op one line "bufs bufe overflow_count"; at the end the ring and its mirrored tail as hex.
static int ring_selftest(int argc, char** argv) {
    if (argc < 5)
        return 2;
    input_t in;
    memset(&in, 0, sizeof(in));
    in.buf_size = (size_t)atol(argv[2]);
    fft_size = (size_t)atol(argv[3]);
    in.bytes_per_sample = atoi(argv[4]);
    const size_t tail = 2 * (size_t)in.bytes_per_sample * fft_size;
    std::vector<unsigned char> store(in.buf_size + tail, 0xEE);
    in.buffer = store.data();
    pthread_mutex_init(&in.buffer_lock, NULL);
    unsigned counter = 0;
    std::vector<unsigned char> scratch, chunk;
    for (int i = 5; i < argc; i++) {
        const char op = argv[i][0];
        const size_t n = (size_t)atol(argv[i] + 1);
        if (op == 'a') {
            chunk.resize(n);
            for (size_t k = 0; k < n; k++)
                chunk[k] = (unsigned char)(counter++ * 7u + 1u);
            circbuffer_append(&in, chunk.data(), n);
        } else if (op == 'c') {
            in.bufs = (in.bufs + n) % in.buf_size;
        } else if (op == 'r') {
            const unsigned char* p = ring_contiguous(&in, n, scratch);
            printf("r ");
            for (size_t k = 0; k < n; k++)
                printf("%02x", p[k]);
            printf("\n");
        }
        printf("%zu %zu %zu %zu\n", in.bufs, in.bufe, in.overflow_count, ring_fill(&in));
    }
    for (unsigned char b : store)
        printf("%02x", b);
    printf("\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "--ring-selftest") == 0)
        return ring_selftest(argc, argv);
    if (argc < 4) {
        fprintf(stderr, "usage: %s <config.txt> <capture.iq> <out_prefix> [gpu]\n", argv[0]);
        return 2;
    }
    const int gpu = argc > 4 ? atoi(argv[4]) : 0;
    FILE* cf = fopen(argv[1], "r");
    if (!cf) {
        perror("config");
        return 2;
    }
    int sample_rate, centerfreq, fftlog, sfmt, tau, quadri;
    if (fscanf(cf, "%d %d %d %d %d %d", &sample_rate, &centerfreq, &fftlog, &sfmt, &tau, &quadri) != 6) {
        fprintf(stderr, "bad device line\n");
        return 2;
    }
    std::vector<mi_channel_cfg> chans;
    while (true) {
        mi_channel_cfg c;
        int n = fscanf(cf, "%d %d %d %d %f %f %f %f %d %f %d %d %d", &c.freq, &c.modulation, &c.squelch_threshold_dbfs, &c.has_snr_threshold,
                       &c.squelch_snr_db, &c.notch_freq, &c.notch_q, &c.ctcss_freq, &c.bandwidth, &c.ampfactor, &c.tau, &c.afc, &c.has_iq_outputs);
        if (n != 13)
            break;
        chans.push_back(c);
    }
    fclose(cf);
    if (chans.empty()) {
        fprintf(stderr, "no channels\n");
        return 2;
    }
    fft_size_log = (size_t)fftlog;
    fft_size = (size_t)1 << fftlog;
    fm_quadri_demod_selected = quadri;

    input_t* in = input_new_for_format((sample_format_t)sfmt, sample_rate, centerfreq);
    device_t* dev = device_new(in, chans.data(), (int)chans.size(), tau);
    devices = dev;
    device_count = 1;
    devices_running = 1;
    Signal sig;
    demod_params_t dp;
    if (init_demod(&dp, &sig, 0, 1, gpu) != 0)
        return 1;  // the reference calls error() here (rtl_airband.cpp:318-332)

    Sink sink;
    for (size_t i = 0; i < chans.size(); i++) {
        std::string p = std::string(argv[3]) + "_ch" + std::to_string(i) + ".f32";
        sink.audio.push_back(fopen(p.c_str(), "wb"));
        sink.flags.emplace_back();
        rawfile_out_t rf;
        if (chans[i].has_iq_outputs)
            rf.f = fopen((std::string(argv[3]) + "_ch" + std::to_string(i) + ".cf32").c_str(), "wb");
        sink.raw.push_back(rf);
        sink.udp.push_back(fopen((std::string(argv[3]) + "_ch" + std::to_string(i) + ".udp").c_str(), "wb"));
    }
    FILE* iq = fopen(argv[2], "rb");
    if (!iq) {
        perror("capture");
        return 2;
    }
    in->state = INPUT_RUNNING;
    pthread_t th;
    pthread_create(&th, NULL, &demodulate, &dp);

    const size_t hop = mi_demod_hop_bytes(dev->engine);
    std::vector<unsigned char> chunk(hop * 1000);  // < one batch
    auto starved = [&]() {
        pthread_mutex_lock(&in->buffer_lock);
        size_t avail = in->bufe >= in->bufs ? in->bufe - in->bufs : in->buf_size - in->bufs + in->bufe;
        pthread_mutex_unlock(&in->buffer_lock);
        return avail < mi_demod_bytes_consumed(dev->engine, 1) + fft_size * (size_t)in->bytes_per_sample * 2;
    };
    auto settle = [&]() {  // wait until the demod thread cannot run and the output side has consumed what it produced
        for (int spin = 0; spin < 200000; spin++) {
            if (output_consume(dev, 0, &sink_fn, &sink))
                sink.batches++;
            if (starved() && !dev->waveavail) {
                usleep(2000);  // the demod thread may be between its availability check and the engine call
                if (starved() && !dev->waveavail)
                    return;
            }
            usleep(200);
        }
    };
    while (true) {
        size_t len = fread(chunk.data(), 1, chunk.size(), iq);
        if (len == 0)
            break;
        circbuffer_append(in, chunk.data(), len);
        settle();
    }
    settle();
    do_exit = 1;
    pthread_join(th, NULL);
    fclose(iq);
    for (FILE* f : sink.audio)
        fclose(f);
    for (rawfile_out_t& r : sink.raw)
        if (r.f)
            fclose(r.f);
    for (FILE* f : sink.udp)
        fclose(f);
    std::string fp = std::string(argv[3]) + "_axc.txt";
    FILE* ff = fopen(fp.c_str(), "w");
    for (const std::string& s : sink.flags)
        fprintf(ff, "%s\n", s.c_str());
    fclose(ff);
    printf("batches=%zu overruns=%zu overflows=%zu\n", sink.batches, dev->output_overrun_count, in->overflow_count);
    device_free(dev);
    input_free(in);
    return 0;
}
