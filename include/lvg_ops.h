/*
 * lvg_ops.h -- C ABI of liblvg_hip.so, the MI355X (gfx950) kernel library behind
 * the torch_utils.ops hot path of LongVideoGAN.
 *
 * Every entry point replaces one pybind function of the reference plugins
 * (paths relative to the reference repo):
 *
 *   lvg_bias_act            <- bias_act()            torch_utils/ops/bias_act.cpp:32
 *   lvg_upfirdn2d           <- upfirdn2d()           torch_utils/ops/upfirdn2d.cpp:16
 *   lvg_filtered_lrelu      <- filtered_lrelu()      torch_utils/ops/filtered_lrelu.cpp:16
 *   lvg_filtered_lrelu_act  <- filtered_lrelu_act_() torch_utils/ops/filtered_lrelu.cpp:213
 *   lvg_tapconv_epilogue[_backward]  (same, with the temporal-tap sum of a tap-stacked convolution)
 *   lvg_conv3d_frames       (the dense contraction itself + the two epilogues above: F.conv3d of
 *                           temporal_modulated_conv3d, model/generator_lres.py:119, which the reference
 *                           hands to cuDNN)
 *   lvg_video_to_uint8 / lvg_video_from_uint8  (utils.py:163 write_video_grid, dataset.py:81 read_frame: tensor code
 *                           in the reference, no plugin)
 *   lvg_modconv_epilogue[_backward]  (no pybind counterpart: fuses the modulated-conv epilogue the
 *                           reference spells in Python, model/generator_lres.py:101-123,570-574)
 *
 * Contract (differs from the pybind ABI on purpose):
 *   - plain pointers and sizes only; the caller owns every buffer (outputs are
 *     allocated by the host language, e.g. torch.empty), nothing is allocated,
 *     freed or synchronised inside the library;
 *   - every launch goes to the hipStream_t passed as `stream` (0 = the null
 *     stream); the library holds no global device state (the reference's global
 *     filter buffers, filtered_lrelu.cu:77-78, are gone), so calls on different
 *     streams may overlap;
 *   - return value: 0 = launched, LVG_ERR_* < 0 = refused, nothing launched.
 *     LVG_ERR_UNSUPPORTED from lvg_filtered_lrelu is the reference's
 *     "return code -1": no fused kernel for these parameters, the caller must
 *     take the generic path (filtered_lrelu.py:223-229);
 *   - lvg_last_error() returns a thread-local message for the last refusal.
 *   - strides are in ELEMENTS, shapes are NCHW order [n, c, h, w].
 */
#ifndef LVG_OPS_H
#define LVG_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVG_ABI_VERSION 1

/* element types of activations; filters are always float32 */
enum { LVG_F32 = 0, LVG_F16 = 1, LVG_BF16 = 2, LVG_F64 = 3 };

/* activation ids: identical to `cuda_idx` of bias_act.activation_funcs (bias_act.py:21-31) */
enum {
    LVG_ACT_LINEAR = 1, LVG_ACT_RELU = 2, LVG_ACT_LRELU = 3, LVG_ACT_TANH = 4, LVG_ACT_SIGMOID = 5,
    LVG_ACT_ELU = 6, LVG_ACT_SELU = 7, LVG_ACT_SOFTPLUS = 8, LVG_ACT_SWISH = 9
};

enum {
    LVG_OK = 0,
    LVG_ERR_INVALID = -1,      /* bad argument (the reference raises via TORCH_CHECK) */
    LVG_ERR_UNSUPPORTED = -2,  /* no specialised kernel: take the generic path       */
    LVG_ERR_LAUNCH = -3        /* hipLaunchKernel / hipGetLastError reported a failure */
};

/* sign-tensor modes of filtered_lrelu */
enum { LVG_SIGNS_NONE = 0, LVG_SIGNS_WRITE = 1, LVG_SIGNS_READ = 2 };

int lvg_abi_version(void);
const char* lvg_last_error(void);

/*
 * y = clamp(act(x + b[(i / stepB) % sizeB]) * gain)                      grad == 0
 * y = x(=dy) * act'(.) * gain,     zeroed where |yref| >= clamp          grad == 1
 * y = x(=d_dx) * act''(.) * gain * dy, same mask                         grad == 2
 * All tensors are dense with identical layout, n elements; b/xref/yref/dy may be NULL.
 * clamp < 0 disables clamping. Reference: bias_act.cpp:32-90, bias_act.cu:23-147.
 */
int lvg_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                 void* y, int64_t n, int64_t sizeB, int64_t stepB, int dtype, int grad, int act,
                 float alpha, float gain, float clamp, void* stream);

/*
 * grad == 1 of lvg_bias_act on a channels-last stream (element i belongs to channel i % channels) that also leaves the
 * bias gradient: db_partial [slots, channels] float32 holds, per workgroup, the sums of the (rounded) dx over the
 * workgroup's pixels; db = sum over the slots (the reference reduces the stored dx in a second pass: bias_act.py:183
 * `dx.sum(...)`). lvg_bias_act_grad_bias_slots returns the number of slots, or 0 when the form does not apply
 * (channels must be a multiple of the 16-byte vector with 256 % (channels / vector) == 0, n >= 1024 vectors, n %
 * channels == 0): use lvg_bias_act + a reduction then. Every activation except swish.
 */
int64_t lvg_bias_act_grad_bias_slots(int64_t n, int channels, int dtype);
int lvg_bias_act_grad_bias(const void* dy, const void* xref, const void* yref, void* dx, float* db_partial,
                           int64_t n, int channels, int dtype, int act, float alpha, float gain, float clamp, void* stream);

/*
 * upsample (zero insertion) -> pad/crop -> FIR -> decimate, per channel.
 * Exactly one of {f2d} or {fx, fy} describes the filter:
 *   f2d != NULL : dense 2-D taps, f2d[fy_i * fstride_y + fx_i * fstride_x], size fh x fw
 *                 (reference plugin form, upfirdn2d.cpp:16);
 *   f2d == NULL : separable; fx (fw taps, may be NULL = identity) along W and
 *                 fy (fh taps, may be NULL = identity) along H, applied as ONE fused
 *                 pass (the reference issues two plugin calls, upfirdn2d.py:241-245).
 * out size = (in*up + pad0 + pad1 - f + down) / down (upfirdn2d.cpp:35-36); the caller
 * passes it in yshape. gain multiplies the result. flip != 0 means correlation.
 */
int lvg_upfirdn2d(const void* x, void* y, const float* f2d, const float* fx, const float* fy,
                  const int64_t xshape[4], const int64_t xstride[4],
                  const int64_t yshape[4], const int64_t ystride[4],
                  int fw, int fh, int64_t fstride_x, int64_t fstride_y,
                  int upx, int upy, int downx, int downy,
                  int padx0, int pady0, int flip, float gain, int dtype, void* stream);

/*
 * Fused bias -> up-FIR (x up^2) -> gain -> leaky ReLU -> clamp -> down-FIR, one pass
 * (filtered_lrelu.cpp:16-209). fu/fd are separable 1-D float32 taps (fu_n / fd_n of
 * them); the 1x1 case is fu_n == fd_n == 1 with up == down == 1.
 * sign_mode WRITE: s receives 2 bits per up-sampled pixel (bit0 negative, bit1 clamped),
 *   4 pixels per byte, rows of sshape[1] bytes, sshape[0] rows, per (n, c) plane;
 * sign_mode READ: slope/zero decisions come from s at offset (sofs_x, sofs_y) instead
 *   of from the data (backward pass, filtered_lrelu.py:239-268).
 * Returns LVG_ERR_UNSUPPORTED when (up, down, taps, dtype) has no fused kernel.
 */
int lvg_filtered_lrelu(const void* x, void* y, const void* b, uint8_t* s,
                       const float* fu, const float* fd,
                       const int64_t xshape[4], const int64_t xstride[4],
                       const int64_t yshape[4], const int64_t ystride[4],
                       int fu_n, int fd_n, int up, int down, int px0, int py0,
                       const int64_t sshape[2], int sofs_x, int sofs_y, int sw_active,
                       float gain, float slope, float clamp, int flip, int sign_mode,
                       int dtype, void* stream);

/* (Which fused kernel serves a call is the library's choice; the test / measurement override lives in lvg_test_hooks.h.) */

/* 1 if lvg_filtered_lrelu has a fused kernel for these parameters, else 0 (no launch). */
int lvg_filtered_lrelu_supported(int fu_n, int fd_n, int up, int down, int dtype);

/*
 * In-place middle step of the generic filtered_lrelu path: x = clamp(lrelu(x * gain))
 * with sign write / read (filtered_lrelu.cpp:213-290, filtered_lrelu.cu:1105-1211).
 * sshape = {bytes per row, rows}; with WRITE the sign plane covers x exactly
 * (width rounded up to 16 pixels).
 */
int lvg_filtered_lrelu_act(void* x, uint8_t* s, const int64_t xshape[4], const int64_t xstride[4],
                           const int64_t sshape[2], int sofs_x, int sofs_y,
                           float gain, float slope, float clamp, int sign_mode,
                           int dtype, void* stream);

/*
 * Epilogue of a style-modulated convolution fused with the prologue of the next one, one pass:
 *   out[f,c,p] = clamp(act(y[f,c,p] * pre[f,c] + b[c]) * gain, +-clamp) * post[f,c]
 *   msq[s, f]  = partial sum of (value before `post`)^2   (optional input-magnitude statistic; see "slots")
 * y/out: dense [frames, channels, pixels] (channels_last = 0) or [frames, pixels, channels]
 * (channels_last = 1) in `dtype` (f32/f16/bf16); pre/post: float32 [frames, channels] or NULL (= 1);
 * b: [channels] in `dtype` or NULL; msq: float32 [slots, frames] or NULL.
 * Reductions are written as PARTIAL sums in `slots` = lvg_modconv_epilogue_slots(...) slices that the caller adds in a
 * fixed order (no atomics: results are reproducible run to run; every element of every slice is written, no zero-fill).
 * act: LVG_ACT_LINEAR / RELU / LRELU (others: LVG_ERR_UNSUPPORTED). clamp < 0 disables clamping.
 * Replaces the reference's Python-level sequence  output * demodulation  (model/generator_lres.py:122),
 * bias_act (:570, torch_utils/ops/bias_act.cpp:32), input * style (:101) and the magnitude
 * statistic (:574) -- there is no single reference entry point; the binding is this library's own.
 */
int lvg_modconv_epilogue(const void* y, const float* pre, const void* b, const float* post, void* out, float* msq,
                         int64_t frames, int channels, int pixels, int channels_last, int dtype, int act,
                         float alpha, float gain, float clamp, void* stream);

/*
 * Backward of lvg_modconv_epilogue with the activation recomputed from y:
 *   du = dout * post * [|g| < clamp] * gain * act'(y*pre + b);   dy = du * pre
 *   d_pre[f,c] = sum_p du * y;  d_post[f,c] = sum_p dout * g;  d_sum[f,c] = sum_p du  (db = sum_f d_sum)
 * d_pre / d_post / d_sum: float32 [slots, frames, channels] partial sums (slots as above with backward = 1; d_pre /
 * d_post may be NULL when pre / post are).
 */
int lvg_modconv_epilogue_backward(const void* dout, const void* y, const float* pre, const void* b, const float* post,
                                  void* dy, float* d_pre, float* d_post, float* d_sum,
                                  int64_t frames, int channels, int pixels, int channels_last, int dtype, int act,
                                  float alpha, float gain, float clamp, void* stream);

/*
 * Dual form of the two functions above for channels-last tensors: the forward pass also writes `mid` = the value before
 * `post` (the activated tensor a skip connection reads, next to the modulated one the convolution reads: the block-final
 * bias_act of model/generator_lres.py:575 and the next block's `input * style` of :101 in ONE pass, 3 streams instead of 4),
 * the backward pass takes the gradients of both outputs: du = (dout * post + dmid) * [inside] * gain * act'(u) (4 streams
 * instead of the 9 of modulate-backward + gradient sum + bias_act-backward). mid / dmid may be NULL.
 */
int lvg_modconv_epilogue_dual(const void* y, const float* pre, const void* b, const float* post, void* out, void* mid, float* msq,
                              int64_t frames, int channels, int pixels, int dtype, int act,
                              float alpha, float gain, float clamp, void* stream);
int lvg_modconv_epilogue_dual_backward(const void* dout, const void* dmid, const void* y, const float* pre, const void* b, const float* post,
                                       void* dy, float* d_pre, float* d_post, float* d_sum,
                                       int64_t frames, int channels, int pixels, int dtype, int act,
                                       float alpha, float gain, float clamp, void* stream);
int lvg_modconv_epilogue_slots(int64_t frames, int channels, int pixels, int channels_last, int dtype, int backward);

/*
 * Temporal-tap gather fused with the epilogue above (channels-last only). A kt x kh x kw convolution over
 * time-major frames is run as ONE 2-D convolution whose output channels stack the kt taps,
 * z [frames, pixels, taps*channels] (tap-major); this entry point performs the temporal sum while applying
 * the epilogue:
 *   ysum[f,p,c] = sum_k z[f + (k - taps/2) * tap_shift, p, k*channels + c]     (frames outside -> 0)
 *   out[f,p,c]  = clamp(act(ysum * pre[f,c] + b[c] + res[f,p,c]) * gain, +-clamp) * post[f,c];  msq[s, f] as above with
 *                 slots = lvg_tapconv_epilogue_slots(...) (also for d_pre / d_post / d_sum of the backward entry point)
 * res (layout of out) and ysum (saved for the backward pass) may be NULL. tap_shift = frames per time step.
 * Replaces, next to the sequence cited for lvg_modconv_epilogue, the accumulation of the per-tap
 * convolution outputs (the reference's conv3d does it inside cuDNN: model/generator_lres.py:119).
 */
int lvg_tapconv_epilogue(const void* z, const float* pre, const void* b, const void* res, const float* post,
                         void* out, void* ysum, float* msq,
                         int64_t frames, int channels, int pixels, int taps, int64_t tap_shift,
                         int dtype, int act, float alpha, float gain, float clamp, void* stream);
int lvg_tapconv_epilogue_slots(int64_t frames, int channels, int pixels, int dtype);

/*
 * Implicit-GEMM convolution on the matrix cores with the temporal-tap sum and the epilogue above fused on
 * store (16-bit channels-last frames; csrc/conv3d_igemm.hip). Replaces F.conv3d of the reference's
 * temporal_modulated_conv3d (model/generator_lres.py:119, padding = k // 2: :544-548) together with
 * lvg_tapconv_epilogue / lvg_modconv_epilogue:
 *   x [frames, H, W, ci] with x_pixel_stride elements between pixels (0 = ci; larger when x is a channel slice of a
 *   wider channels-last tensor); w [kt, kh, kw, co, ci] (tap-major, input channel fastest)
 *   acc[f,h,v,o] = sum_{dt,dh,dw,c} x[f + (dt-kt/2)*frame_shift, h+dh-kh/2, v+dw-kw/2, c] * w[dt,dh,dw,o,c]   (zero outside)
 *   out = clamp(act(acc * pre[f,o] + b[o] + res[f,h,v,o]) * gain, +-clamp) * post[f,o];  ysum = acc (may be NULL)
 *   msq_partial[i] = sum over workgroup i of (value before post)^2, i < lvg_conv3d_frames_workgroups(...)
 *                    (fixed summation order: reproducible; the caller adds them up; may be NULL)
 * Returns LVG_ERR_UNSUPPORTED when no kernel exists for the shape (ci % 64, co % 64, kt <= 7, kh * kw <= 25, odd kernel
 * sizes, frames*h*w < 2^31): the caller then takes the library convolution + lvg_tapconv_epilogue.
 */
/* lvg_conv3d_frames with a choice of output type: out_dtype = dtype (the call below), or LVG_F32 -- out, ysum, b and res are then float32
 * tensors and the float32 accumulators are stored unrounded: the output side of a float32-accurate contraction from operands split into
 * 16-bit high / low parts stacked along ci (long-video-gan_amd/torch_utils/ops/conv3d_frames.py, `split32`). The reference runs these
 * convolutions in float32 with TF32 off (train_lres.py:267-269). lvg_conv3d_frames_workgroups_f32out: its workgroup count. */
int lvg_conv3d_frames_ex(const void* x, const void* w, const float* pre, const void* b, const void* res, const float* post,
                         void* out, void* ysum, float* msq_partial,
                         int64_t frames, int h, int wd, int ci, int co, int kt, int kh, int kw, int64_t frame_shift,
                         int64_t x_pixel_stride, int dtype, int out_dtype, int act, float alpha, float gain, float clamp, void* stream);
int64_t lvg_conv3d_frames_workgroups_f32out(int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw);
int lvg_conv3d_frames(const void* x, const void* w, const float* pre, const void* b, const void* res, const float* post,
                      void* out, void* ysum, float* msq_partial,
                      int64_t frames, int h, int wd, int ci, int co, int kt, int kh, int kw, int64_t frame_shift,
                      int64_t x_pixel_stride, int dtype, int act, float alpha, float gain, float clamp, void* stream);

/*
 * Weight gradient of lvg_conv3d_frames (csrc/conv3d_wgrad.hip; what autograd derives for the reference's F.conv3d,
 * model/generator_lres.py:119, discriminator_lres.py:169): 3 x 3 spatial taps, kt <= 7 temporal taps, 16-bit
 * channels-last frames whose width is 8 / 16 / 32 / 64, ci and co multiples of 64.
 *   part[s, dt, dh*3+dw, o, c] = sum over the pixels of range s of dy[f,h,v,o] * x[f + (dt-kt/2)*frame_shift, h+dh-1, v+dw-1, c]
 * part: float32 [splits, kt, 9, co, ci] with splits = lvg_conv3d_frames_wgrad_splits(...) (0 = no kernel for the
 * shape); the caller adds the ranges (fixed order: reproducible). zeros: >= 128 zero bytes in device memory.
 * x / dy may be channel slices of wider channels-last tensors (pixel strides in elements, 0 = dense).
 */
int lvg_conv3d_frames_wgrad(const void* x, const void* dy, float* part, const void* zeros,
                            int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw, int64_t frame_shift,
                            int64_t x_pixel_stride, int64_t dy_pixel_stride, int splits, int dtype, void* stream);
int lvg_conv3d_frames_wgrad_splits(int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw);

/* Tiles lvg_conv3d_frames cuts this shape into (= length of msq_partial; one workgroup per tile, or a persistent grid walking them);
 * 0 = unsupported shape. */
int64_t lvg_conv3d_frames_workgroups(int64_t frames, int h, int wd, int ci, int co, int kt, int kh, int kw);


/*
 * Backward of lvg_tapconv_epilogue from dout and the saved ysum; the gradient is written already scattered
 * into the tap-stacked layout, dz[f + (k - taps/2) * tap_shift, p, k*channels + c] = dy[f,p,c] (zeros where
 * the source frame lies outside), every element of dz exactly once. d_pre / d_post / d_sum as above.
 */
int lvg_tapconv_epilogue_backward(const void* dout, const void* ysum, const float* pre, const void* b, const void* res,
                                  const float* post, void* dz, float* d_pre, float* d_post, float* d_sum,
                                  int64_t frames, int channels, int pixels, int taps, int64_t tap_shift,
                                  int dtype, int act, float alpha, float gain, float clamp, void* stream);

/*
 * Prologue / epilogue of the 2-D style-modulated convolution of the super-resolution generator, fused with the
 * NCHW <-> NHWC layout change (csrc/modconv2d_layout.hip). float16 / bfloat16 only; p = y * W + x.
 *
 * lvg_modconv2d_nchw_to_nhwc:  dst[n, p, c] = src[n, c, p] * scale[n, c], src = src_a (c_a channels) followed by src_b
 *   (c_b channels, may be NULL / 0); channels c_a + c_b .. c_dst - 1 of dst are zero (c_dst % 8 == 0). scale: float32
 *   [n, c_a + c_b] or NULL. With oth (NHWC [n, p, c_oth], c_oth % 8 == 0) and partial (float32 [n, ceil(hw / 64), c_a + c_b]):
 *   partial[n, t, c] = sum over the 64 pixels of tile t of src[n, c, p] * oth[n, p, c]  (sum over t = the gradient of scale
 *   of the inverse transform; no atomics, fixed summation order).
 * lvg_modconv2d_nhwc_to_nchw:  dst[n, c, p] = src[n, p, c] * scale[n, c] for c < c_dst (src has c_src >= c_dst channels,
 *   c_src % 8 == 0; scale float32 [n, c_dst] or NULL). With oth_a / oth_b (NCHW, c_a + c_b <= c_src channels) and partial
 *   (float32 [n, ceil(hw / 64), c_a + c_b]): partial[n, t, c] = sum over tile t of src[n, p, c] * oth[n, c, p].
 *
 * Replace the reference's Python-level  x * styles  /  x * dcoefs  around the grouped convolution of
 * modulated_conv2d (model/generator_sres.py:24-67; there the style is folded into per-sample weights) and the
 * torch.cat with the conditioning frames (model/generator_sres.py:463). No single reference entry point: the
 * binding is this library's own (long-video-gan_amd/torch_utils/ops/modconv2d_layout.py).
 */
int lvg_modconv2d_nchw_to_nhwc(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                               int64_t n, int64_t hw, int c_a, int c_b, int c_dst, int c_oth, int dtype, void* stream);
int lvg_modconv2d_nhwc_to_nchw(const void* src, const float* scale, const void* oth_a, const void* oth_b, void* dst, float* partial,
                               int64_t n, int64_t hw, int c_src, int c_dst, int c_a, int c_b, int dtype, void* stream);
/* lvg_modconv2d_nchw_to_nhwc writing into the interior of a LARGER channels-last frame: source planes src_h x src_w go to
 * dst [n][dst_h][dst_w][c_dst] at (off_y, off_x) (the explicit zero padding of lvg_conv2d_frames / lvg_conv2d_frames_wgrad).
 * zero_border != 0: the border pixels are zero-filled by the call (dst may be uninitialised memory); 0: the border is not written
 * (the caller zero-filled dst). oth / partial as above (dense frames of src_h * src_w pixels). */
int lvg_modconv2d_nchw_to_nhwc_padded(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                                      int64_t n, int src_h, int src_w, int c_a, int c_b, int c_dst, int c_oth,
                                      int dst_h, int dst_w, int off_y, int off_x, int zero_border, int dtype, void* stream);
/* ... with a planar reduction partner oth [n][c_oth][src_h][src_w] (the source's own layout; c_oth <= c_a + c_b): partial = per-tile sums of src * oth. */
int lvg_modconv2d_nchw_to_nhwc_padded_planar(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                                      int64_t n, int src_h, int src_w, int c_a, int c_b, int c_dst, int c_oth,
                                      int dst_h, int dst_w, int off_y, int off_x, int zero_border, int dtype, void* stream);

/*
 * Dense 3 x 3 contraction of the super-resolution networks as a hand-written implicit GEMM (csrc/conv2d_igemm.hip) and its
 * weight gradient (csrc/conv2d_wgrad.hip), on channels-last frames whose zero padding is written explicitly in memory
 * ('valid' correlation; float16 / bfloat16, float32 accumulation):
 *
 *   lvg_conv2d_frames:  out[n][oy][ox][co] = pre[n][co] * sum_{dh, dw, ci} x[n][oy + in_off_y + dh][ox + in_off_x + dw][ci] * w[dh][dw][co][ci]
 *     x [n][hi][wi] pixels of x_pixel_stride elements (>= ci, % 8; 0 = ci), w [3][3][co][ci] tap-major, out [n][ho][wo] pixels of
 *     out_pixel_stride elements (0 = co), ho <= hi - in_off_y - 2, wo <= wi - in_off_x - 2; pre float32 [n][co] or NULL (= 1); ci % 64 == 0, co % 64 == 0;
 *     fewer than 2^32 bytes of x. Run on the padded output gradient with the weight mirrored in both taps and its channel
 *     roles exchanged it is the data gradient. out_dtype = dtype, or LVG_F32: the float32 accumulators are stored unrounded (the
 *     output side of a float32-accurate contraction from operands split into 16-bit high / low parts stacked along ci).
 *   lvg_conv2d_frames_wgrad:  part[s][dh][dw][co][ci] = sum over the K-steps of range s of dy[n][oy][ox][co] * x[n][oy + dh][ox + dw][ci]
 *     dy [n][hd][wd] (hd % 4 == 0, wd % 16 == 0: 4 x 16 pixel patches; rows / columns past the true gradient hold zeros),
 *     x [n][hx][wx] with hx >= hd + 2, wx >= wd + 2 (finite everywhere); part float32 [splits][3][3][co][ci], the caller adds the
 *     ranges in order (reproducible); splits = lvg_conv2d_frames_wgrad_splits(...) (0: no kernel for the shape).
 *   lvg_conv2d_frames_workgroups: workgroups lvg_conv2d_frames launches (0: no kernel for the shape).
 *
 * Replace the library convolution inside the reference's modulated_conv2d (model/generator_sres.py:63-66:
 * conv2d_gradfix.conv2d(..., padding = kernel - 1, groups = batch) -- here one dense convolution between the style / demodulation
 * passes of lvg_modconv2d_*) and what autograd derives for it (torch_utils/ops/conv2d_gradfix.py:37-45). No reference plugin
 * entry point (the reference calls cuDNN through F.conv2d); binding: long-video-gan_amd/torch_utils/ops/conv2d_frames.py.
 */
int64_t lvg_conv2d_frames_workgroups(int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int kh, int kw);
int lvg_conv2d_frames(const void* x, const void* w, const float* pre, void* out,
                      int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int kh, int kw, int in_off_y, int in_off_x,
                      int64_t x_pixel_stride, int64_t out_pixel_stride, int dtype, int out_dtype, void* stream);
/* ... with the result stored as NCHW planes for a consumer that tiles planes (filtered_lrelu): out [n][co_out][ho][wo] = pre[n][co_out] * acc for the first
 * co_out <= co channels (co: the padded channel count of w; pre may be NULL), x's dtype; wo even. Replaces lvg_conv2d_frames + lvg_modconv2d_nhwc_to_nchw
 * (the demodulation `x * dcoefs` of model/generator_sres.py:54-55, applied to the output). */
int lvg_conv2d_frames_planes(const void* x, const void* w, const float* pre, void* out,
                             int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int co_out, int kh, int kw, int in_off_y, int in_off_x,
                             int64_t x_pixel_stride, int dtype, void* stream);
/* ... and, from the same accumulators (before `pre`), dot_partial[n][row][c] = sum over the pixels of half tile `row` of acc * oth[n][c][pixel] for
 * c < c_dot_a + c_dot_b <= co; oth = dot_a [n][c_dot_a][ho][wo] and dot_b [n][c_dot_b][ho][wo] concatenated along the channels (x's dtype, 4-byte aligned; dot_b may
 * be NULL with c_dot_b = 0); rows per frame = lvg_conv2d_frames_planes_dot_rows(ho, wo); the caller adds the rows in order (reproducible). The data gradient of the
 * modulated convolution: dx * styles leaves as planes, d styles = sum dx * x comes from the accumulators (model/generator_sres.py:61 differentiated). */
int64_t lvg_conv2d_frames_planes_dot_rows(int ho, int wo);
int lvg_conv2d_frames_planes_dot(const void* x, const void* w, const float* pre, void* out, const void* dot_a, const void* dot_b, float* dot_partial,
                                 int c_dot_a, int c_dot_b,
                                 int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int co_out, int kh, int kw, int in_off_y, int in_off_x,
                                 int64_t x_pixel_stride, int dtype, void* stream);
int lvg_conv2d_frames_wgrad_splits(int64_t n, int hx, int wx, int hd, int wd, int ci, int co, int kh, int kw);
int lvg_conv2d_frames_wgrad(const void* x, const void* dy, float* part,
                            int64_t n, int hx, int wx, int hd, int wd, int ci, int co, int kh, int kw,
                            int64_t x_pixel_stride, int64_t dy_pixel_stride, int splits, int dtype, void* stream);

/*
 * Adam step over one flat float32 range, optionally followed by the exponential moving average of the updated
 * weights, in one pass (csrc/optim.hip):
 *   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 *   p_ema += (p - p_ema) * ema_weight            (p_ema may be NULL)
 * torch.optim.Adam's arithmetic (amsgrad off, no weight decay); all ranges 16-byte aligned, `step` >= 1 is the
 * update count of this range. Replaces the reference's torch.optim.Adam.step() (model/video_gan_lres.py:83-90, used
 * at :132, :176, :203) and the lerp of update_G_ema (:208-214); no single reference entry point -- the binding is
 * this library's own (long-video-gan_amd/lvg/optim.py).
 */
int lvg_adam_step(float* p, const float* g, float* m, float* v, float* p_ema, int64_t n,
                  float lr, float beta1, float beta2, float eps, int64_t step, float ema_weight, void* stream);

/*
 * Weight side of a modulated convolution in one pass per direction (csrc/weight_prep.hip): the per-output-channel max
 * normalisation, the 1 / sqrt(fan_in) scale, the sum of squares over the taps (demodulation term) and the cast to the
 * compute dtype of model/generator_lres.py:97-119, written in the [taps, co, ci] layout lvg_conv3d_frames consumes.
 *   forward:  w [co, ci, taps] f32 -> wp [taps, co, ci] (dtype), w2 [co, ci] f32 (may be NULL), amax [co] f32
 *   backward: g = gradient of wp's elements addressed as g[co*s0 + ci*s1 + tap*s2] (dtype; g_strides = {s0, s1, s2} in
 *             elements), g_w2 [co, ci] f32 or NULL -> dw [co, ci, taps] f32. Ties in max|w| share the gradient.
 * ci * taps * 8 bytes must fit in LDS (150 KiB) and ci <= 1024.
 */
int lvg_weight_prep(const float* w, void* wp, float* w2, float* amax, int co, int ci, int taps, float scale, int normalize,
                    int dtype, void* stream);
int lvg_weight_prep_backward(const float* w, const float* amax, const void* g, const int64_t* g_strides, const float* g_w2,
                             float* dw, int co, int ci, int taps, float scale, int normalize, int dtype, void* stream);
/* wp [taps, co, ci] (16-bit elements) -> wt [taps, ci, co] with the tap order reversed: the weight of the data-gradient
 * convolution (mirrored taps, channel roles swapped: what lvg_conv3d_frames takes as `w` when it is run on dy). */
int lvg_weight_dgrad_pack(const void* wp, void* wt, int taps, int co, int ci, void* stream);

/* out[plane] = sum of the hw elements of plane `plane` of a contiguous [planes][hw] tensor (float32 accumulation, fixed order):
 * the per-(sample, channel) part of the bias gradient dx.sum([0, 2, 3]) of filtered_lrelu's backward pass
 * (reference torch_utils/ops/filtered_lrelu.py:254); the caller adds the samples of a channel. */
int lvg_plane_sum(const void* x, float* out, int64_t planes, int64_t hw, int dtype, void* stream);

/* out[plane] = sum of the SQUARES of the hw elements of plane `plane` (float32 accumulation, fixed order): one pass over a 16-bit or
 * float32 activation for the mean-square statistic of the generators' input-magnitude EMAs, x.float().square().mean() in the reference
 * (model/generator_sres.py:278-286, model/generator_lres.py:298-312): the caller adds the planes and divides by the element count. */
int lvg_plane_sum_sq(const void* x, float* out, int64_t planes, int64_t hw, int dtype, void* stream);
/* out[plane] = max |x| over the plane (NaN-ignoring fmax): the tensor maximum behind the power-of-two scale of a split-precision operand
 * (torch_utils/ops/conv2d_frames.py::pow2_scale) without an abs() copy of the tensor. */
int lvg_plane_absmax(const void* x, float* out, int64_t planes, int64_t hw, int dtype, void* stream);

/*
 * Operand of the split-precision contraction of the sres generator's float32 layers in one pass (no reference counterpart: the reference
 * runs these layers as float32 convolutions, model/generator_sres.py:24-67): float32 NCHW planes src [n, c, src_h, src_w], times mul[n, c]
 * (or NULL), times the device scalar scale[0] (or NULL; a power of two), split into float16 parts (part 0 = rounding of the value,
 * part 1 = rounding of the remainder) and written channels-last into the interior (off_y, off_x) of a frame dst [n, dst_h, dst_w,
 * n_blocks * c_pad] as n_blocks stacked channel blocks; bit blk of `pattern` selects the part of block blk. Border and padding channels are
 * left untouched (zero-fill the frame once).
 */
int lvg_split16_frames(const float* src, const float* mul, const float* scale, void* dst, int64_t n, int c, int src_h, int src_w,
                       int dst_h, int dst_w, int off_y, int off_x, int c_pad, int n_blocks, int pattern, void* stream);
/* The float32 result of that contraction back to NCHW planes: dst[n, c, p] = src[n, p, c] * scale[n, c] * factor[0] for c < c_dst (src has
 * c_src >= c_dst channels per pixel; scale / factor may be NULL). */
int lvg_nhwc_f32_to_nchw(const float* src, const float* scale, const float* factor, float* dst, int64_t n, int64_t hw, int c_src, int c_dst, void* stream);

/*
 * Fused stages of the ADA augmentation pipeline (csrc/ada_augment.hip; reference model/ada_augment.py).
 *   lvg_ada_warp: the geometric stage (:271-304: reflect padding by `margins`, x2 up-sampling with the 12-tap low-pass, bilinear
 *     resampling through the inverse affine map (affine_grid + grid_sample, zeros outside, align_corners = False), x2 down-sampling
 *     with the flipped filter) in one launch. x, y [n][k][h][w] float32 (k = channels x frames of a sample), g_inv [n][3][3] the map in
 *     centred pixel units BEFORE the padding / over-sampling adjustments of :287-296, margins int32[4] = (mx0, my0, mx1, my1) ON THE
 *     DEVICE (the rule of :275-284; the reference reads them back to the host), taps [12] the normalised 1-D filter.
 *   lvg_ada_colour: y = C[:3, :3] . x + C[:3, 3] (cmat [n][4][4] or NULL), + noise * sigma[n] (noise like x, or NULL), zero inside the
 *     cutout rectangle |(px + 0.5) / w - cx| < sx / 2 and |(py + 0.5) / h - cy| < sy / 2 (cut [n][4] = cx, cy, sx, sy, or NULL) on
 *     x [n][3][t][h][w] float32 (:376-381, :407-427). transpose bit 0: the backward pass (d x = C[:3, :3]^T . (d y where kept));
 *     bit 1: without the offset column C[:3, 3] (the linear part alone: the backward of the backward).
 */
int lvg_ada_warp(const float* x, const float* g_inv, const int* margins, const float* taps, float* y, int n, int k, int h, int w, void* stream);
/* d x = A^T d y for the linear map y = A x of lvg_ada_warp, by gathers only (no atomics, fixed summation order): the transposed
 * down-sampler through lvg_upfirdn2d into `workspace` (n * k * (h + 6) * 2 * (w + 6) * 2 floats), then one kernel for the transposed
 * bilinear sampling (through the inverse map), the transposed up-sampler and the reflection folding. */
int lvg_ada_warp_adjoint(const float* dy, const float* g_inv, const int* margins, const float* taps, float* workspace, float* dx,
                         int n, int k, int h, int w, void* stream);
int lvg_ada_colour(const float* x, const float* cmat, const float* noise, const float* sigma, const float* cut, float* y,
                   int n, int t, int h, int w, int transpose, void* stream);

/*
 * Weight side of the 2-D modulated convolution of the super-resolution generator (reference model/generator_sres.py:50-58 and :63,
 * `weight.to(x.dtype)`) in one pass per direction (csrc/weight_prep.hip):
 *   forward:  w [co, ci, taps] f32 -> w' = w * rsqrt(mean over (ci, taps) of w^2) * scale;
 *             wp [taps, co_pad, ci_pad] = w' in `dtype`, padding zero-filled (the weight lvg_conv2d_frames consumes);
 *             wt [taps, ci_pad, co_pad] = wp with the taps mirrored and the channel roles exchanged (the data-gradient weight), or NULL;
 *             w2 [co, ci] f32 = sum over the taps of w'^2;  stat [co] f32 = rsqrt(mean w^2) (kept for the backward pass)
 *   backward: g = gradient of wp's elements, FLOAT32 [taps, co_pad, ci_pad] (what lvg_conv2d_frames_wgrad produces) or NULL,
 *             g_w2 [co, ci] f32 or NULL (at least one of the two) -> dw [co, ci, taps] f32. Linear in (g, g_w2): the two
 *             contributions may be taken in separate calls and added.
 * ci * taps * 8 bytes must fit in LDS (150 KiB).
 */
int lvg_weight_prep2d(const float* w, void* wp, void* wt, float* w2, float* stat, int co, int ci, int taps, int co_pad, int ci_pad,
                      float scale, int dtype, void* stream);
int lvg_weight_prep2d_backward(const float* w, const float* stat, const float* g, const float* g_w2, float* dw,
                               int co, int ci, int taps, int co_pad, int ci_pad, float scale, void* stream);

/*
 * Style side of a modulated convolution (csrc/style_prep.hip): the per-sample max normalisation of the styles and the
 * demodulation term of model/generator_lres.py:99 and :107-108, on styles in frames order s [t, n, ci] float32 (row
 * r = t_index * n + n_index), with w2 [co, ci] = sum over the taps of the squared scaled weight (lvg_weight_prep).
 *   forward:  mod [t*n, ci] = s / amax[n_index],  amax [n] = max |s| over (t, ci),
 *             demod [t*n, co] = rsqrt(sum_ci w2[co, ci] * mod[r, ci]^2 + 1e-8)
 *   backward: g_mod [t*n, ci] (or NULL), g_demod [t*n, co] -> ds [t, n, ci], dw2 [co, ci]; gm_scratch [t*n, ci] is
 *             work space. Ties in max|s| share the gradient (torch.amax). Reductions run in a fixed order.
 * ci and co multiples of 4, t * n < 2^24; all tensors float32, dense, 16-byte aligned.
 */
int lvg_style_prep(const float* s, const float* w2, float* mod, float* demod, float* amax, int t, int n, int ci, int co, void* stream);
int lvg_style_prep_backward(const float* s, const float* amax, const float* w2, const float* mod, const float* demod,
                            const float* g_mod, const float* g_demod, float* gm_scratch, float* ds, float* dw2,
                            int t, int n, int ci, int co, void* stream);

/*
 * The two ends of the pixel path, one HBM pass each (csrc/video_io.hip).
 *   lvg_video_to_uint8:   video [n, c, t, h, w] (dtype) -> bytes [n, t, h, w, c] uint8 = (x * 127.5 + 128).clamp(0, 255)
 *                         truncated: utils.py:163 / :203 (write_video_grid / save_image_grid) + the channel-last
 *                         rearrangement of :171 / :209, bit-identical for float32 input.
 *   lvg_video_from_uint8: bytes [n, t, h, w, c] uint8 -> video [n, c, t, h, w] (dtype) = 2 * float(x) / 255 - 1
 *                         (dataset.py:81-83 read_frame), mirrored in x for samples with flip[i] != 0 (:93-94 x_flip;
 *                         flip = device pointer to n bytes, or NULL).
 * 1 <= c <= 4, w a multiple of 4; dense tensors.
 */
int lvg_video_to_uint8(const void* video, void* bytes, int64_t n, int c, int t, int h, int w, int dtype, void* stream);
int lvg_video_from_uint8(const void* bytes, void* video, const uint8_t* flip, int64_t n, int c, int t, int h, int w, int dtype, void* stream);

/*
 * Temporal noise filter bank of the low-resolution generator (csrc/noise_bank.hip). Replaces the grouped F.conv1d of
 * model/generator_lres.py:378-388 (BlurredNoise.blur: every noise row against each of the F right-aligned low-pass
 * filters of K taps, then the per-filter scale of :383-384):
 *   out [rows, filters, frames] = scale[f] * sum_k noise[r, t + k] * bank[f, k],   length = frames + taps - 1.
 * The bank is passed PACKED per group of 32 filters (built once per bank by the caller): group g walks only its last
 * 2 * pairs[g] taps (pairs[g] = ceil(longest filter of the group / 2) rounded up to a multiple of 64), and
 *   bankP[(pairOff[g] + p) * 64 + lane] = bank[32 g + lane % 32][taps - 2 pairs[g] + 2 p + lane / 32]   (0 outside the bank),
 * pairOff [groups + 1] int32 = prefix sums of pairs[], maxPairs = the largest pairs[g]; bankP ends in 8 spare zero lines
 * of 64 floats (the kernel prefetches one block ahead). Float32 MFMA with float32
 * accumulation; the K range of a tile is split over four waves and summed in a fixed order (reproducible). All pointers
 * are device pointers, float32, dense; scale may be NULL.
 */
int lvg_noise_filter_bank(const float* noise, const float* bankP, const int* pairOff, const float* scale, float* out,
                          int rows, int length, int frames, int filters, int taps, int groups, int maxPairs, void* stream);

/*
 * 1 x 1 convolutions with a THIN side of 1 .. 4 channels on channels-last 16-bit frames (csrc/pointwise_thin.hip): the
 * generator's ToRGB (model/generator_lres.py:600-640: a [3, C, 1, 1, 1] weight) and the discriminator's first layer
 * (model/discriminator_lres.py:169, Conv3dLayer(3, 32, 1, 1)) with their gradients -- HBM streams over the wide tensor.
 * pixels = frames * H * W; wide = 8 / 16 / 32 / 64 / 128 channels; w float32 [thin, wide]; float32 accumulation.
 *   lvg_pointwise_thin_out:   y [pixels, thin] = x [pixels, wide] . w^T         (x 16-byte aligned)
 *   lvg_pointwise_thin_in:    y [pixels, wide] = x [pixels, thin] . w           (y 16-byte aligned)
 *   lvg_pointwise_thin_wgrad: partial [blocks, thin, wide] float32, sum over blocks = thinT^T . wideT  (wideT [pixels, wide],
 *                             thinT [pixels, thin]); blocks = lvg_pointwise_thin_wgrad_blocks(pixels, wide); every workgroup
 *                             owns a contiguous pixel range and reduces in a fixed order (reproducible).
 */
int lvg_pointwise_thin_out(const void* x, const float* w, void* y, int64_t pixels, int wide, int thin, int dtype, void* stream);
int lvg_pointwise_thin_in(const void* x, const float* w, void* y, int64_t pixels, int wide, int thin, int dtype, void* stream);
int lvg_pointwise_thin_wgrad_blocks(int64_t pixels, int wide);
int lvg_pointwise_thin_wgrad(const void* wideT, const void* thinT, float* partial, int64_t pixels, int wide, int thin, int dtype, int blocks, void* stream);

/*
 * Weight gradient of a 1 x 1 convolution on channels-last 16-bit frames (csrc/pointwise_wgrad.hip): what autograd derives for the skip
 * convolutions of model/generator_lres.py:558-577 and model/discriminator_lres.py:169 (kernel size 1),
 *   part [splits, co, ci] float32, sum over splits = sum over pixels dy[m][co] * x[m][ci]
 * on the matrix cores with the pixel index as K (both operands staged as they lie, transpose reads). ci, co multiples of 64, pixels a
 * multiple of 8; x_pixel_stride / dy_pixel_stride = elements between consecutive pixels (0: dense; channel slices and pixel-pair views
 * are fine), multiples of 8; zeros = 128 bytes of zeros in device memory; splits = lvg_pointwise_wgrad_splits(pixels, ci, co)
 * (0 = unsupported). Every split owns a contiguous pixel range; the caller adds the splits in order (reproducible).
 */
int lvg_pointwise_wgrad_splits(int64_t pixels, int ci, int co);
int lvg_pointwise_wgrad(const void* x, const void* dy, float* part, const void* zeros, int64_t pixels, int ci, int co,
                        int64_t x_pixel_stride, int64_t dy_pixel_stride, int splits, int dtype, void* stream);

/*
 * Operand preparation of the float32 route (csrc/split32.hip): the reference trains the low-resolution networks in float32 with TF32 off
 * (train_lres.py:267-269); here float32 tensors run on the 16-bit matrix cores as three bfloat16 parts t = t1 + t2 + t3 (t1 = bf16(t),
 * t2 = bf16(t - t1), t3 = bf16(t - t1 - t2), round to nearest even). One pass over x [pixels, channels] float32 (x_pixel_stride elements
 * between pixels, 0 = dense) writes out [pixels, blocks * channels] bfloat16 whose channel block k holds part (pattern >> 2 k) & 3:
 * pattern 0x910 (blocks 6: parts 0 0 1 0 1 2) for a convolution input, 0x24 (blocks 3: parts 0 1 2) for the weight-gradient operands.
 * channels a multiple of 8; pointers 16-byte aligned.
 */
int lvg_split32_stack(const float* x, void* out, int64_t pixels, int channels, int64_t x_pixel_stride, int blocks, int pattern, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LVG_OPS_H */
