"""`torch_utils.ops.conv2d_resample`: 2-D convolution combined with FIR up / down-sampling
(reference torch_utils/ops/conv2d_resample.py:46-141). The dense contraction goes to
`conv2d_gradfix` (MIOpen, MFMA), every resampling filter to the HIP `upfirdn2d`. Filter and
convolution are ordered per case so that the contraction always runs at the LOWER resolution:

    1x1 weight, down     decimate (filter)        -> convolve
    1x1 weight, up       convolve                 -> interpolate (filter)
    k x k,     down      blur at full resolution  -> strided convolution
    any,       up        transposed strided conv  -> blur (-> decimate)
    no resampling        plain convolution (symmetric padding) or explicit pad -> convolve"""

import torch

from .. import misc
from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


def _get_weight_shape(w):
    with misc.suppress_tracer_warnings():  # constant under tracing
        shape = [int(sz) for sz in w.shape]
    misc.assert_shape(w, shape)
    return shape


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d / conv_transpose2d. `flip_weight=True` means correlation (what the library computes),
    False mirrors the taps first, i.e. a true convolution."""
    kh, kw = _get_weight_shape(w)[2:]
    if (kh > 1 or kw > 1) and not flip_weight:
        w = w.flip([2, 3])
    run = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return run(x, w, stride=stride, padding=padding, groups=groups)


def _delay_compensated_padding(padding, fw, fh, up, down):
    """User padding plus the group delay of the resampling filter (same rule as upsample2d / downsample2d)."""
    pad = list(_parse_padding(padding))                     # x0, x1, y0, y1
    for axis, taps in ((0, fw), (2, fh)):
        if up > 1:
            pad[axis] += (taps + up - 1) // 2
            pad[axis + 1] += (taps - up) // 2
        if down > 1:
            pad[axis] += (taps - down + 1) // 2
            pad[axis + 1] += (taps - down) // 2
    return pad


def _swap_in_out_channels(w, groups):
    """Weight layout conv_transpose2d expects: [Cin, Cout/groups, kh, kw]."""
    if groups == 1:
        return w.transpose(0, 1)
    cout, cin_g, kh, kw = _get_weight_shape(w)
    w = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2)
    return w.reshape(groups * cin_g, cout // groups, kh, kw)


@misc.profiled_function
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    r"""x: [N, Cin, H, W]; w: [Cout, Cin//groups, kh, kw] in x's dtype; f: filter from
    `upfirdn2d.setup_filter()` or None; up / down: integer factors; padding (int, [x, y] or
    [x0, x1, y0, y1]) is measured on the up-sampled image."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32)
    for name, value in (('up', up), ('down', down), ('groups', groups)):
        assert isinstance(value, int) and value >= 1, name
    kh, kw = _get_weight_shape(w)[2:]
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _delay_compensated_padding(padding, fw, fh, up, down)
    fir = dict(f=f, flip_filter=flip_filter)
    conv = dict(w=w, groups=groups, flip_weight=flip_weight)
    pointwise = kh == 1 and kw == 1

    if up == 1 and down > 1:
        if pointwise:       # a 1x1 contraction commutes with the filter: decimate first
            x = upfirdn2d.upfirdn2d(x=x, down=down, padding=[px0, px1, py0, py1], **fir)
            return _conv2d_wrapper(x=x, **conv)
        x = upfirdn2d.upfirdn2d(x=x, padding=[px0, px1, py0, py1], **fir)
        return _conv2d_wrapper(x=x, stride=down, **conv)

    if up > 1:
        if pointwise and down == 1:
            x = _conv2d_wrapper(x=x, **conv)
            return upfirdn2d.upfirdn2d(x=x, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, **fir)
        # the transposed strided convolution inserts the zeros; what it cannot pad away is left to the filter
        px0, px1 = px0 - (kw - 1), px1 - (kw - up)
        py0, py1 = py0 - (kh - 1), py1 - (kh - up)
        crop_x, crop_y = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=_swap_in_out_channels(w, groups), stride=up, padding=[crop_y, crop_x], groups=groups,
                            transpose=True, flip_weight=not flip_weight)
        x = upfirdn2d.upfirdn2d(x=x, padding=[px0 + crop_x, px1 + crop_x, py0 + crop_y, py1 + crop_y], gain=up ** 2, **fir)
        return upfirdn2d.upfirdn2d(x=x, down=down, **fir) if down > 1 else x

    # up == down == 1
    if px0 == px1 >= 0 and py0 == py1 >= 0:
        return _conv2d_wrapper(x=x, padding=[py0, px0], **conv)
    x = upfirdn2d.upfirdn2d(x=x, f=None, padding=[px0, px1, py0, py1], flip_filter=flip_filter)    # pad / crop only
    return _conv2d_wrapper(x=x, **conv)
