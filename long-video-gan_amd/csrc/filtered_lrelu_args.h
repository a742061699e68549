// filtered_lrelu_args.h -- launch arguments shared by the two fused filtered_lrelu kernels of liblvg_hip.so:
// the fp32-VALU kernel (filtered_lrelu.hip: float32 I/O, exact float32 intermediates) and the MFMA kernel
// (filtered_lrelu_mfma.hip: float16 / bfloat16 I/O, the four FIR stages as banded matrix products).
#pragma once
#include "lvg_common.h"

struct FlreluArgs
{
    const void*  x;
    void*        y;
    const void*  b;
    uint8_t*     s;
    const float* fu;
    const float* fd;
    int64_t      xs[4], ys[4];
    int          n, c, xh, xw, yh, yw;
    int          fuN, fdN;       // actual tap counts (<= template FU / FD)
    int          px0, py0;
    int          sWBytes, sH;    // mask plane: bytes per row, rows
    int          sOfsX, sOfsY;
    int          swLimit;        // bytes per row that carry pixels
    float        gain, slope, clamp;
    int          flip;
    int          tilesX, tilesY;
    int64_t      xLoB, xHiB;     // bytes of the x tensor below / above p.x: every address in [x - xLoB, x + xHiB) belongs to it
};

enum { LVG_FLRELU_CFG_NONE = 0, LVG_FLRELU_CFG_POINTWISE, LVG_FLRELU_CFG_U2D2, LVG_FLRELU_CFG_U4D2, LVG_FLRELU_CFG_U2D4 };

// filtered_lrelu_mfma.hip. dtype is LVG_F16 or LVG_BF16; cfg one of U2D2 / U4D2 / U2D4.
int lvg_flrelu_mfma_launch(FlreluArgs& p, int cfg, int mode, int dtype, hipStream_t stream);
// filtered_lrelu_wave.hip (round 4: one wave per tile, no workgroup barrier). Same arguments.
int lvg_flrelu_wave_launch(FlreluArgs& p, int cfg, int mode, int dtype, hipStream_t stream);
// filtered_lrelu_band.hip (round 5: one workgroup per plane, rows through an LDS ring by LDS-DMA, streaming vertical stages; float16 only).
// LVG_ERR_UNSUPPORTED for what it does not take: the caller falls back to the wave kernel. Same arguments.
// `all` = 0: only the shapes it measured faster on than the wave kernel (else LVG_ERR_UNSUPPORTED); 1: everything it can take.
int lvg_flrelu_band_launch(FlreluArgs& p, int cfg, int mode, int dtype, int all, hipStream_t stream);
// filtered_lrelu_strip.hip (round 6: one wave per 24-column strip, wave-private LDS ring, no barrier; float16 only).
// LVG_ERR_UNSUPPORTED for what it does not take. `all` = 0: only the shapes it measured faster on; 1: everything it can take.
int lvg_flrelu_strip_launch(FlreluArgs& p, int cfg, int mode, int dtype, int all, hipStream_t stream);
