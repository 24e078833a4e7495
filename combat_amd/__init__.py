"""combat_amd -- MI355X-native (gfx950) implementation of COMBAT's alternated
generator/surrogate training step (reference: train_generator.py:170-290).

Layout: ``csrc/`` HIP kernels behind the C ABI declared in ``include/combat_hip.h``;
``_lib`` ctypes binding; ``engine`` per-network forward/backward schedules over NHWC bf16
buffers; ``nets`` reference-compatible parameter containers; ``step`` the alternated step;
``trigger``/``augment``/``data``/``dist``/``log`` host logic around it.
"""
__version__ = "0.1.0"
