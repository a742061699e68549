"""Sampling pipeline (reference generate.py:56-88, generator_lres.py:778-816, generator_sres.py:662-681): low-resolution
video -> super-resolution in segments -> display bytes.

What is different from the reference, by design:
* the low-resolution generator can STREAM: it is convolutional in time with a fixed receptive field (temporal_padding
  frames per level), so frames [t0, t0 + C) of a video of length L (L and C multiples of the total temporal scale, 32)
  are what the generator produces from the slice [t0, t0 + C + margin) of the temporal embedding -- up to the per-chunk
  statistics the networks take over time (the style max-normalisation runs over the chunk's frames, so the 1e-8 of the
  demodulation and 16-bit roundings differ per chunk: 1e-5 in float32, tests/test_pixel_path.py). The reference runs the
  whole clip at once (activation memory grows with the video length); `lres_video_chunks` keeps it bounded.
  Streaming needs a clip length that is a multiple of 32; the reference's own lengths (ceil(len / 16) * 16 [+ 2 * context]
  = 304 / 312 for the default 301 frames) are not, and such clips run in ONE pass exactly like the reference's. With
  `pad_to_scale=True` the clip is lengthened to the next multiple of 32 and cropped: bounded memory for any length, but the
  embedding is then drawn for the longer clip, i.e. a given seed no longer reproduces the reference's frames;
* every segment leaves the device as uint8 [N, T, H, W, C] (`lvg.video_io.video_to_uint8`, one pass) -- the layout the
  reference's writer builds per frame on the host (utils.py:163-171).
mp4 muxing (imageio-ffmpeg in the reference) stays outside: `generate_video` yields the frame bytes."""

from typing import Iterator, Optional

import torch

from . import video_io


def lres_video_chunks(G, temporal_emb: torch.Tensor, seq_length: int, chunk: int = 128,
                      dtype: Optional[torch.dtype] = None) -> Iterator[torch.Tensor]:
    """Frames of `G.forward_from_emb(temporal_emb, seq_length)` in pieces of `chunk` frames, [N, 3, chunk, H, W] float32,
    each computed from its own slice of the embedding (bounded activation memory)."""
    scale = G.total_temporal_scale
    assert seq_length % scale == 0 and chunk % scale == 0 and chunk > 0, f'streaming needs lengths that are multiples of {scale}'
    margin = temporal_emb.shape[2] - seq_length              # receptive-field margin of the embedding (2 * padding * scale)
    for t0 in range(0, seq_length, chunk):
        n = min(chunk, seq_length - t0)
        yield G.forward_from_emb(temporal_emb[:, :, t0:t0 + n + margin].contiguous(), n, dtype=dtype)


def lres_video(G, batch_size: int, seq_length: int, generator_emb: Optional[torch.Generator] = None, chunk: Optional[int] = 128,
               dtype: Optional[torch.dtype] = None, pad_to_scale: bool = False) -> Iterator[torch.Tensor]:
    """Low-resolution video as an iterator over time pieces. Streams when the length is a multiple of the total temporal scale,
    else one piece -- the embedding is drawn once, exactly as `G(batch_size, seq_length, ...)` draws it. `pad_to_scale`: generate
    the next multiple of the scale instead (always streams; the last piece is cropped; NOT the reference's frames for a seed)."""
    scale = G.total_temporal_scale
    if pad_to_scale and chunk and seq_length % scale:
        padded = (seq_length + scale - 1) // scale * scale
        emitted = 0
        for piece in lres_video(G, batch_size, padded, generator_emb, chunk, dtype):
            piece = piece[:, :, :seq_length - emitted]
            emitted += piece.shape[2]
            if piece.shape[2]:
                yield piece
        return
    emb = G.sample_temporal_emb(batch_size, seq_length, generator_emb)
    if chunk and seq_length % scale == 0 and seq_length > chunk:
        yield from lres_video_chunks(G, emb, seq_length, max(scale, chunk // scale * scale), dtype)
    else:
        yield G.forward_from_emb(emb, seq_length, dtype=dtype)


@torch.no_grad()
def generate_video(lres_G, sres_G=None, seq_length: int = 301, seed: Optional[int] = None, batch_size: int = 1,
                   segment_length: int = 16, lres_chunk: Optional[int] = 128, dtype: Optional[torch.dtype] = None,
                   as_uint8: bool = True, return_lres: bool = False, pad_to_scale: bool = False) -> Iterator:
    """Frames of `seq_length`-frame videos, segment by segment: uint8 [N, T_seg, H, W, 3] (or float [N, 3, T_seg, H, W]
    with as_uint8=False); with `return_lres` each item is (high-res segment, matching low-res segment).

    Lengths and random streams follow reference generate.py: the low-resolution clip is ceil(len / 16) * 16 frames plus
    2 * temporal_context when a super-resolution network is given; ONE torch.Generator seeded with `seed` drives the
    temporal embedding and then the super-resolution latent."""
    device = next(lres_G.parameters()).device
    out = (lambda v: video_io.video_to_uint8(v)) if as_uint8 else (lambda v: v)
    lr_len = (seq_length + segment_length - 1) // segment_length * segment_length
    ctx = sres_G.temporal_context if sres_G is not None else 0
    generator = None if seed is None else torch.Generator(device).manual_seed(seed)
    pieces = lres_video(lres_G, batch_size, lr_len + 2 * ctx, generator, lres_chunk, dtype, pad_to_scale)
    emitted = 0
    if sres_G is None:
        for piece in pieces:
            piece = piece[:, :, :max(0, seq_length - emitted)]
            emitted += piece.shape[2]
            if piece.shape[2]:
                yield out(piece)
        return
    latent_z = None
    window = None                                            # low-res frames not yet consumed (with their left context)
    for piece in pieces:
        window = piece if window is None else torch.cat([window, piece], dim=2)
        if latent_z is None:                                 # drawn after the embedding, like the reference's call order ...
            latent_z = sres_G.sample_latent_z(batch_size, generator)
        while window.shape[2] >= segment_length + 2 * ctx and emitted < seq_length:
            lr_seg = window[:, :, :segment_length + 2 * ctx]
            hr = sres_G.SG3(latent_z, lr_seg)
            keep = min(segment_length, seq_length - emitted)
            emitted += keep
            item = out(hr[:, :, :keep])
            yield (item, out(lr_seg[:, :, ctx:ctx + keep])) if return_lres else item
            window = window[:, :, segment_length:]
