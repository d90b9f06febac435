"""vision_ft_amd — MI355X-native hot path of p1atdev/vision-ft.

Host side: a Python mirror of the reference's operator interface (``src/modules``, ``src/models``,
``src/config.py``, ``src/trainer``) whose heavy operators are hand-written HIP kernels for gfx950,
reached through the C ABI in ``include/vft_hip.h`` (``libvft_hip.so``).  No CPU fallback.
"""

__version__ = "0.1.0"
