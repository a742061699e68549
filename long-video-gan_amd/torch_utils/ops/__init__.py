"""torch_utils.ops -- the reference's custom-op API (bias_act, upfirdn2d, filtered_lrelu,
conv2d_gradfix, conv2d_resample, fma, grid_sample_gradfix) backed by hand-written HIP kernels
for MI355X (gfx950) through the C ABI of liblvg_hip.so (include/lvg_ops.h)."""
