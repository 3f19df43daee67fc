"""svg — MI355X-native drop-in for the attention-processor / operator API of svg-project/Sparse-VideoGen.

Module paths, class names and function signatures mirror the reference package `svg` so that the reference's entry
scripts (`from svg.models.hyvideo.inference import replace_hyvideo_attention`, ...) work unchanged; the sparse hot
path underneath is libsvgattn.so (hand-written HIP for gfx950, see include/svg_attn.h).
"""
__all__ = ["_native"]
