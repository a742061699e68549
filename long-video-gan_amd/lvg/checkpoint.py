"""Checkpoint I/O of the training-step bodies (reference train_lres.py:165-178, video_gan_lres.py:218-233 `ckpt()`):
the generator EMA alone (what `generate` loads) and the full training state -- G, D, both optimizers, step.

The reference pickles module OBJECTS through `torch_utils.persistence` (the pickle embeds the model source). This
repo's networks are state-dict compatible re-implementations, so a checkpoint here is (constructor kwargs, state dict)
per network + the flat Adam moments, written with torch.save; reference-written pickles still load through
`torch_utils.persistence` (tests/test_persistence.py) and their state dicts load into these networks."""

from typing import Optional

import torch


def _module_state(module) -> Optional[dict]:
    if module is None:
        return None
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _own_rng(trainer) -> dict:
    rng = dict(cpu=torch.get_rng_state())
    dev = getattr(trainer, 'device', None)
    if dev is not None and torch.device(dev).type == 'cuda' and torch.cuda.is_available():
        rng['cuda'] = torch.cuda.get_rng_state(torch.device(dev))
    return rng


def trainer_state(trainer, step: int, all_ranks: bool = False) -> dict:
    """Everything `load_trainer_state` needs to resume `trainer`: networks, generator EMA, both optimizers, the random streams (CPU and
    the trainer's GPU) and -- for the super-resolution trainer -- the ADA state the reference's `ckpt()` stores next to the networks
    (model/video_gan_sres.py: `augment`, `in_augment`, `real_sign_collector`): the adapted probability `augment.p`, the conditioning-side
    pipe's buffers and the real-sign statistics accumulated since the last probability update.

    Ranks hold identical copies of the networks and optimizers, but NOT of the random streams (every rank draws its own latents,
    augmentations and noise: train_lres.py:69). `rng` is therefore the CALLING rank's stream, tagged with its rank and the world size, and
    is only restored into that same rank of a run of the same size. `all_ranks=True` (a collective: every rank must call) additionally
    gathers every rank's streams into `rng_ranks`, so that a multi-rank resume continues every stream bit for bit."""
    rank, world = _rank_world()
    state = dict(format='lvg-train-1', step=int(step),
                 G=_module_state(trainer.G), D=_module_state(trainer.D), G_ema=_module_state(getattr(trainer, 'G_ema', None)),
                 G_opt=_cpu(trainer.G_opt.state_dict()), D_opt=_cpu(trainer.D_opt.state_dict()),
                 rng=_own_rng(trainer), rng_rank=rank, rng_world=world)
    if all_ranks and world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, _cpu(state['rng']))
        state['rng_ranks'] = gathered
    for name in ('augment', 'in_augment'):
        if getattr(trainer, name, None) is not None:
            state[name] = _module_state(getattr(trainer, name))
    if hasattr(trainer, '_real_sign_sum'):
        state['real_sign_sum'] = trainer._real_sign_sum.detach().cpu().clone()
    return state


def _cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


def save_checkpoint(path, trainer, step: int) -> None:
    torch.save(trainer_state(trainer, step), path)


def save_G_ema(path, trainer, init_kwargs: Optional[dict] = None) -> None:
    """The inference artefact: generator EMA weights + the constructor arguments to rebuild the network (`init_kwargs`, default: the
    arguments the trainer built its generator with, `trainer.G_init_kwargs`)."""
    net = trainer.G_ema if getattr(trainer, 'G_ema', None) is not None else trainer.G
    if init_kwargs is None:
        init_kwargs = getattr(trainer, 'G_init_kwargs', None)
        assert init_kwargs is not None, 'save_G_ema: pass the generator constructor arguments (the trainer does not record them)'
    torch.save(dict(format='lvg-G-1', init_kwargs=dict(init_kwargs), state=_module_state(net)), path)


def load_trainer_state(trainer, state: dict) -> int:
    """Restore `trainer` in place from `trainer_state` / `save_checkpoint` output; returns the step to continue from."""
    assert state.get('format') == 'lvg-train-1', 'not a training checkpoint of this repo'
    with torch.no_grad():
        trainer.G.load_state_dict(state['G'])
        trainer.D.load_state_dict(state['D'])
        if state.get('G_ema') is not None and getattr(trainer, 'G_ema', None) is not None:
            trainer.G_ema.load_state_dict(state['G_ema'])
        for name in ('augment', 'in_augment'):
            if state.get(name) is not None:
                assert getattr(trainer, name, None) is not None, f'the checkpoint holds `{name}` state but the trainer was built without it'
                getattr(trainer, name).load_state_dict(state[name])
        if state.get('real_sign_sum') is not None and hasattr(trainer, '_real_sign_sum'):
            trainer._real_sign_sum.copy_(state['real_sign_sum'])
    trainer.G_opt.load_state_dict(state['G_opt'])
    trainer.D_opt.load_state_dict(state['D_opt'])
    # Random streams: a rank continues ITS OWN stream or keeps the one it has. Restoring the saving rank's stream into every rank would
    # make all ranks draw identical latents / augmentations / noise from then on.
    rank, world = _rank_world()
    rng = None
    if state.get('rng_ranks') is not None and len(state['rng_ranks']) == world:
        rng = state['rng_ranks'][rank]
    elif 'rng' in state and state.get('rng_world', 1) == world and state.get('rng_rank', 0) == rank:
        rng = state['rng']
    if rng is not None:
        if 'cpu' in rng:
            torch.set_rng_state(rng['cpu'])
        if 'cuda' in rng and torch.cuda.is_available() and torch.device(getattr(trainer, 'device', 'cpu')).type == 'cuda':
            torch.cuda.set_rng_state(rng['cuda'], torch.device(trainer.device))
    return int(state['step'])


def load_checkpoint(path, trainer) -> int:
    return load_trainer_state(trainer, torch.load(path, map_location='cpu', weights_only=False))


def load_G(path, factory):
    """Rebuild a generator saved by `save_G_ema`: `factory(**init_kwargs)` -> module, weights loaded, eval mode, no grads
    (reference utils.load_G, utils.py:53-56)."""
    blob = torch.load(path, map_location='cpu', weights_only=False)
    assert blob.get('format') == 'lvg-G-1'
    net = factory(**blob['init_kwargs'])
    net.load_state_dict(blob['state'])
    return net.requires_grad_(False).eval()
