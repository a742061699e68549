/*
 * lvg_test_hooks.h -- test / measurement controls of liblvg_hip.so. NOT part of the drop-in operator ABI (include/lvg_ops.h): nothing a
 * maintainer of the reference binds. They exist so that tests can force every implementation of an operator through the same C entry
 * point (tests/test_filtered_lrelu_gpu.py, tests/test_conv3d_frames.py: "every form computes the same bits") and measurements can A/B
 * them inside one process (tools/flrelu_check.cpp, tools/conv_bench.py). Both are process-wide host state; the operators themselves
 * keep no device state (lvg_ops.h, contract).
 */
#ifndef LVG_TEST_HOOKS_H
#define LVG_TEST_HOOKS_H

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Which fused kernel serves float16 / bfloat16 tensors: 0 = default (float16 planes of two / three column strips: the row-band MFMA
 * kernel, csrc/filtered_lrelu_band.hip, LVG_FLRELU_BAND=0 switches it off; everything else: the wave-per-tile MFMA kernel,
 * csrc/filtered_lrelu_wave.hip; LVG_FLRELU_WAVE=0: the round-2 MFMA kernel, csrc/filtered_lrelu_mfma.hip; LVG_FLRELU_MFMA=0: the VALU
 * kernel), 1 = the fp32-VALU kernel (csrc/filtered_lrelu.hip, the only one for float32), 2 = the round-2 MFMA kernel, 3 = the
 * wave-per-tile kernel, 4 = the row-band kernel for everything it can take (falls back to 3), 5 = the strip kernel (round 6,
 * csrc/filtered_lrelu_strip.hip; by default it serves float16 planes of up to four 24-column strips, LVG_FLRELU_STRIP=0 switches it off)
 * for everything it can take (falls back likewise).
 * PROCESS-WIDE (a backward pass launches from autograd's thread, which has to see what the test's thread set); returns the previous
 * setting. Change it only while no other thread is inside lvg_filtered_lrelu.
 */
int lvg_filtered_lrelu_set_impl(int impl);

/* Measurement / test control (no counterpart in the reference): force the tile of lvg_conv3d_frames -- bm pixels (128 | 256) x bn output
 * channels (64 | 128), weight ring depth nb (2 | 3); 0 = the kernel's own choice -- and switch the persistent-workgroup form of the
 * 64-channel tiles on / off. Every form computes the same bits (tests/test_conv3d_frames.py). Initial values: LVG_CONV_BM / _BN / _NB /
 * _PERSIST from the environment. PROCESS-WIDE, read by lvg_conv3d_frames and lvg_conv3d_frames_workgroups (a caller sizes msq_partial
 * with one and launches with the other): change it only while no other thread is inside the library. */
int lvg_conv3d_frames_set_plan(int bm, int bn, int nb, int persist);

#ifdef __cplusplus
}
#endif

#endif
