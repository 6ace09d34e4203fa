// Ablation probe of the fp32 MFMA GEMM main loop (manual tool, see tests/probe/run_probe.py; not shipped in the product .so)
#define CHAM_GEMM_PROBE 1
#include "../../chameleon_recsys_amd/csrc/gemm.hip"
