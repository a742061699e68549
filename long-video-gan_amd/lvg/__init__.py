"""lvg -- MI355X-side pieces that sit next to the drop-in `torch_utils` / `dnnlib` packages:
data-parallel gradient exchange over RCCL (`lvg.ddp`), the low-resolution and super-resolution networks on
the HIP op stack (`lvg.models.lres`, `lvg.models.sres`), DiffAugment / temporal-scale augmentation
(`lvg.augment`), the video ADA pipeline (`lvg.ada_augment`) and the training-step bodies
(`lvg.train_lres`, `lvg.train_sres`; the former is what bench.py times)."""
