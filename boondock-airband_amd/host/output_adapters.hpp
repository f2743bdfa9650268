// output_adapters.hpp -- the byte formats the reference's output thread derives from the demod contract (SURVEY 8f
// rank 4).  Only the payloads are rebuilt, so that a replayed capture yields the same bytes; file naming / rotation,
// sockets, LAME and icecast stay with the reference.
//   rawfile     src/output.cpp:513-561   cf32: one batch = 2 * sizeof(float) * WAVE_BATCH bytes of channel->iq_out
//   udp_stream  src/udp_stream.cpp:86-102, output.cpp:565-576   float32 PCM, mono or L/R interleaved, one datagram per batch
#pragma once
#include <cstddef>
#include <cstdio>

#include "airband_host.hpp"

// O_RAWFILE state that matters for the payload stream: file_data::continuous and output_t::active
struct rawfile_out_t {
    FILE* f = nullptr;
    bool continuous = false;  // file_data::continuous (config key `continuous`)
    bool active = false;      // output_t::active: the previous batch had a signal (output.cpp:560)
    size_t batches_written = 0;
};

// One batch of the output thread's O_RAWFILE branch.  A non-continuous output skips a NO_SIGNAL batch only once the
// previous one was NO_SIGNAL too (output.cpp:516-519), i.e. every transmission is followed by one trailing batch.
// Returns 1 if the batch was written, 0 if skipped, -1 on a short write (the reference disables the output).
int rawfile_put(rawfile_out_t* out, const float* iq_out, char axcindicate);

// O_UDP_STREAM: is a datagram sent for this batch (output.cpp:568-570)?
bool udp_stream_sends(bool continuous, char axcindicate);
// Datagram payloads.  `out` must hold udp_payload_bytes(stereo) bytes; returns the payload length.
size_t udp_payload_bytes(bool stereo);
size_t udp_payload_mono(const float* waveout, unsigned char* out);                             // udp_stream.cpp:86-91
size_t udp_payload_stereo(const float* waveout, const float* waveout_r, unsigned char* out);  // udp_stream.cpp:93-102
