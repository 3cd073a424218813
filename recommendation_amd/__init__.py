"""MI355X-native hot path of wuchaochen/recommendation: embedding gather/pooling, sorted
sparse backward with fused optimizers, and the feature-interaction layers, as hand-written
HIP kernels behind the reference's Python model API."""
__all__ = ['ops']
