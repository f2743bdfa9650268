// l64_args.h -- launch arguments of the lane-resident stage-1 kernel (l64_kernel.h).  Plain C types only: the same text is
// compiled ahead of time by hipcc (channelize_l64.hip) and at run time by hipRTC for a plan's own pruning masks (l64_jit.cpp).
#ifndef MI_L64_ARGS_H
#define MI_L64_ARGS_H

// per channel: the row of its class { bin mod 64 } in the exchange buffer and the twiddles of stages 7, 8, 9 that lead to
// its bin (the sign of an upper output folded in: a - w b == a + (-w) b bit for bit)
struct L64Chan {
    int slot, iq_row;
    float w7x, w7y, w8x, w8y, w9x, w9y;
};

struct L64Args {
    const unsigned char* iq;            // stream s at iq + s * stream_stride
    unsigned long long stream_stride;
    unsigned long long valid_bytes;     // readable bytes per stream starting at its base
    unsigned nfft;                      // windows per stream in this launch
    unsigned plane_off;                 // plane index of window 0
    float* mag;                         // [nstreams * nch][plane_stride]
    float* cplx;                        // [nstreams * n_iq_rows][plane_stride] of (re, im)
    unsigned long long plane_stride;
    const float* window;                // 512 coefficients
    const float* levels;                // 256-entry level table (u8 / s8)
    float conv_scale;                   // 1 / fullscale (s16 / f32)
    int nch, n_iq_rows;
    unsigned* xmax;                     // [nstreams * nch] running max of the magnitudes written (bit pattern), or null
    const L64Chan* chan;                // [nch]
    int nb_pad;                         // channels per window padded to a power of two, 8 .. 64
    unsigned zstride;                   // bytes per window in the exchange buffer
    unsigned span_bytes;                // bytes of the float span of a tile (rows of HOP samples, padded)
    unsigned ntiles;                    // tiles of 32 windows per stream
    unsigned nstreams;
    int sfmt;                           // MI_SFMT_*
    int linear_tiles;                   // (unused)
    unsigned* ticket;                   // zero before the launch: runs of tiles are handed out through it
    unsigned run_tiles;                 // tiles per run
};

#endif
