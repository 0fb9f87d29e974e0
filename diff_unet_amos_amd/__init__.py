"""MI355X-native hot path of Diff-UNet (3D diffusion segmentation).

Host side mirrors the reference's Python interface for this path
(``DiffUNet.forward(image, x, step, pred_type)``, ``SpacedDiffusion``), the
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI in
``include/dua_hip.h`` (``libdua_hip.so``).  No CPU fallback exists.
"""
