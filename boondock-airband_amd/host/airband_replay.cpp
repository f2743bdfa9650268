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
// usage: airband_replay <config.txt>[,<config2.txt>...] <capture.iq>[,<capture2.iq>...] <out_prefix> [gpu]
//   one device per capture; devices with equal configs are the streams of ONE engine (init_demod), served by one submit / wait pair
//   per turn from page-locked rings -- the reference's one-thread-many-devices loop (rtl_airband.cpp:300-306, 381-422)
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

static std::vector<std::string> split_commas(const std::string& s) {
    std::vector<std::string> out;
    size_t from = 0;
    while (true) {
        const size_t c = s.find(',', from);
        out.push_back(s.substr(from, c == std::string::npos ? std::string::npos : c - from));
        if (c == std::string::npos)
            break;
        from = c + 1;
    }
    return out;
}

struct ReplayConfig {
    int sample_rate, centerfreq, fftlog, sfmt, tau, quadri;
    std::vector<mi_channel_cfg> chans;
};

static bool read_config(const char* path, ReplayConfig& c) {
    FILE* cf = fopen(path, "r");
    if (!cf) {
        perror("config");
        return false;
    }
    if (fscanf(cf, "%d %d %d %d %d %d", &c.sample_rate, &c.centerfreq, &c.fftlog, &c.sfmt, &c.tau, &c.quadri) != 6) {
        fprintf(stderr, "bad device line\n");
        fclose(cf);
        return false;
    }
    while (true) {
        mi_channel_cfg ch;
        int n = fscanf(cf, "%d %d %d %d %f %f %f %f %d %f %d %d %d", &ch.freq, &ch.modulation, &ch.squelch_threshold_dbfs, &ch.has_snr_threshold,
                       &ch.squelch_snr_db, &ch.notch_freq, &ch.notch_q, &ch.ctcss_freq, &ch.bandwidth, &ch.ampfactor, &ch.tau, &ch.afc, &ch.has_iq_outputs);
        if (n != 13)
            break;
        c.chans.push_back(ch);
    }
    fclose(cf);
    if (c.chans.empty()) {
        fprintf(stderr, "no channels\n");
        return false;
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "--ring-selftest") == 0)
        return ring_selftest(argc, argv);
    if (argc < 4) {
        fprintf(stderr, "usage: %s <config.txt>[,<config2.txt>...] <capture.iq>[,<capture2.iq>...] <out_prefix> [gpu]\n"
                        "  one device per capture (the last config serves the remaining captures); devices with equal configs share one engine;\n"
                        "  with several devices the output files are <out_prefix>_d<k>_ch<i>.* and <out_prefix>_d<k>_axc.txt\n", argv[0]);
        return 2;
    }
    const int gpu = argc > 4 ? atoi(argv[4]) : 0;
    const std::vector<std::string> cfg_paths = split_commas(argv[1]), cap_paths = split_commas(argv[2]);
    const size_t ndev = cap_paths.size();
    std::vector<ReplayConfig> cfgs(cfg_paths.size());
    for (size_t i = 0; i < cfg_paths.size(); i++)
        if (!read_config(cfg_paths[i].c_str(), cfgs[i]))
            return 2;
    auto cfg_of = [&](size_t d) -> const ReplayConfig& { return cfgs[d < cfgs.size() ? d : cfgs.size() - 1]; };
    fft_size_log = (size_t)cfgs[0].fftlog;  // one fft_size per process, as in the reference (rtl_airband.cpp:808-822)
    fft_size = (size_t)1 << cfgs[0].fftlog;
    fm_quadri_demod_selected = cfgs[0].quadri;

    std::vector<input_t*> ins(ndev);
    devices = (device_t*)calloc(ndev, sizeof(device_t));
    for (size_t d = 0; d < ndev; d++) {
        const ReplayConfig& c = cfg_of(d);
        // (several devices: page-locked rings, read in place by the engine's uploads; one device keeps the plain ring of the reference)
        ins[d] = ndev > 1 ? input_new_pinned_for_format((sample_format_t)c.sfmt, c.sample_rate, c.centerfreq)
                          : input_new_for_format((sample_format_t)c.sfmt, c.sample_rate, c.centerfreq);
        device_t* dev = device_new(ins[d], c.chans.data(), (int)c.chans.size(), c.tau);
        devices[d] = *dev;
        free(dev);  // (the struct was copied into the array; its channels / freqlists live on)
    }
    device_count = (int)ndev;
    devices_running = (int)ndev;
    Signal sig;
    demod_params_t dp;
    if (init_demod(&dp, &sig, 0, (int)ndev, gpu) != 0)
        return 1;  // the reference calls error() here (rtl_airband.cpp:318-332)
    int engines = 0;
    for (size_t d = 0; d < ndev; d++)
        engines += devices[d].engine_owner;

    std::vector<Sink> sinks(ndev);
    auto name = [&](size_t d, const std::string& rest) {
        return std::string(argv[3]) + (ndev > 1 ? "_d" + std::to_string(d) : std::string()) + rest;
    };
    for (size_t d = 0; d < ndev; d++) {
        const ReplayConfig& c = cfg_of(d);
        for (size_t i = 0; i < c.chans.size(); i++) {
            sinks[d].audio.push_back(fopen(name(d, "_ch" + std::to_string(i) + ".f32").c_str(), "wb"));
            sinks[d].flags.emplace_back();
            rawfile_out_t rf;
            if (c.chans[i].has_iq_outputs)
                rf.f = fopen(name(d, "_ch" + std::to_string(i) + ".cf32").c_str(), "wb");
            sinks[d].raw.push_back(rf);
            sinks[d].udp.push_back(fopen(name(d, "_ch" + std::to_string(i) + ".udp").c_str(), "wb"));
        }
    }
    std::vector<FILE*> iq(ndev);
    for (size_t d = 0; d < ndev; d++) {
        iq[d] = fopen(cap_paths[d].c_str(), "rb");
        if (!iq[d]) {
            perror("capture");
            return 2;
        }
        ins[d]->state = INPUT_RUNNING;
    }
    pthread_t th;
    pthread_create(&th, NULL, &demodulate, &dp);

    auto starved = [&](size_t d) {
        input_t* in = ins[d];
        pthread_mutex_lock(&in->buffer_lock);
        size_t avail = in->bufe >= in->bufs ? in->bufe - in->bufs : in->buf_size - in->bufs + in->bufe;
        pthread_mutex_unlock(&in->buffer_lock);
        return avail < mi_demod_bytes_consumed(devices[d].engine, 1) + fft_size * (size_t)in->bytes_per_sample * 2;
    };
    // the demod thread cannot run (every engine has a device whose ring is short -- its devices advance in step) and the output
    // side has consumed what it produced
    auto idle = [&]() {
        for (size_t d = 0; d < ndev; d++)
            if (devices[d].waveavail)
                return false;
        for (size_t d = 0; d < ndev; d++) {
            if (!devices[d].engine_owner)
                continue;
            bool any_short = false;
            for (size_t e = 0; e < ndev; e++)
                if (devices[e].engine == devices[d].engine && starved(e))
                    any_short = true;
            if (!any_short)
                return false;
        }
        return true;
    };
    auto settle = [&]() {
        for (int spin = 0; spin < 200000; spin++) {
            for (size_t d = 0; d < ndev; d++)
                if (output_consume(devices + d, (int)d, &sink_fn, &sinks[d]))
                    sinks[d].batches++;
            if (idle()) {
                usleep(2000);  // the demod thread may be between its availability check and the engine call
                if (idle())
                    return;
            }
            usleep(200);
        }
    };
    std::vector<unsigned char> chunk;
    while (true) {
        size_t fed = 0;
        for (size_t d = 0; d < ndev; d++) {
            chunk.resize(mi_demod_hop_bytes(devices[d].engine) * 1000);  // < one batch
            const size_t len = fread(chunk.data(), 1, chunk.size(), iq[d]);
            if (len)
                circbuffer_append(ins[d], chunk.data(), len);
            fed += len;
        }
        if (!fed)
            break;
        settle();
    }
    settle();
    do_exit = 1;
    pthread_join(th, NULL);
    size_t overruns = 0, overflows = 0;
    for (size_t d = 0; d < ndev; d++) {
        fclose(iq[d]);
        for (FILE* f : sinks[d].audio)
            fclose(f);
        for (rawfile_out_t& r : sinks[d].raw)
            if (r.f)
                fclose(r.f);
        for (FILE* f : sinks[d].udp)
            fclose(f);
        FILE* ff = fopen(name(d, "_axc.txt").c_str(), "w");
        for (const std::string& s : sinks[d].flags)
            fprintf(ff, "%s\n", s.c_str());
        fclose(ff);
        overruns += devices[d].output_overrun_count;
        overflows += ins[d]->overflow_count;
    }
    if (ndev == 1) {
        printf("batches=%zu overruns=%zu overflows=%zu\n", sinks[0].batches, overruns, overflows);
    } else {
        printf("devices=%zu engines=%d batches=", ndev, engines);
        for (size_t d = 0; d < ndev; d++)
            printf("%s%zu", d ? "," : "", sinks[d].batches);
        printf(" overruns=%zu overflows=%zu\n", overruns, overflows);
    }
    for (size_t d = 0; d < ndev; d++) {
        if (devices[d].engine && devices[d].engine_owner)
            mi_demod_destroy(devices[d].engine);
        for (int i = 0; i < devices[d].channel_count; i++)
            free(devices[d].channels[i].freqlist);
        free(devices[d].channels);
        input_free(ins[d]);
    }
    free(devices);
    return 0;
}
