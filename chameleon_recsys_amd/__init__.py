"""chameleon_recsys_amd - MI355X-native NAR (next-article recommendation) training step of CHAMELEON.

Host side mirrors the reference's ``nar_module`` Estimator surface (``nar.*``); the compute path is the
hand-written HIP library ``libchameleon_nar.so`` (C ABI in ``include/chameleon_nar.h``).
"""
__version__ = "0.1.0"
