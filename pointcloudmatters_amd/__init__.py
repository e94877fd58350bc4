"""pointcloudmatters_amd -- MI355X-native behaviour-cloning training path for point-cloud policies.

Sub-packages
------------
pointops   drop-in for the reference's ``pointops`` package (HIP kernels behind a C ABI)
policy     PointNet tokenizer, set-abstraction layer, ACT / Diffusion-Policy heads
bc         the training step (LightningModule.training_step semantics) and data-parallel harness
"""
__version__ = "0.1.0"
