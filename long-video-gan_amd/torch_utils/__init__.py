"""torch_utils -- MI355X-native drop-in for the reference package of the same name.

Only the hot path is provided: `torch_utils.ops.*` (HIP kernels behind the reference's Python
signatures), `persistence` (pickle protocol v6), and the small `misc` / `distributed` /
`training_stats` / `custom_ops` surface that pickled model source imports."""
