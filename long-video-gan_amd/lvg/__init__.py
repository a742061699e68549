"""lvg -- MI355X-side pieces that sit next to the drop-in `torch_utils` / `dnnlib` packages:
data-parallel gradient exchange over RCCL (`lvg.ddp`), the low-resolution networks on the HIP op
stack (`lvg.models.lres`) and the training-step body used by bench.py (`lvg.train_lres`)."""
