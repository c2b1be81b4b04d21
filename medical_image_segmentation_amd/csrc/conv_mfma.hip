// placeholder until the matrix-core kernels land
#include "conv_internal.h"
bool conv3x3_mfma_supported(int, int, int, int, int, int) { return false; }
size_t conv3x3_mfma_workspace_bytes(int, int, int, int, int, int) { return 0; }
int conv3x3_mfma_fwd(const void*, const void*, const float*, int, int, int, int, int, int, void*, float*, void*, hipStream_t) { return MIS_ERR_UNSUPPORTED; }
bool conv3x3_wgrad_mfma_supported(int, int, int, int, int, int) { return false; }
size_t conv3x3_wgrad_mfma_workspace_bytes(int, int, int, int, int, int) { return 0; }
int conv3x3_wgrad_mfma(const void*, const void*, int, int, int, int, int, int, float*, void*, hipStream_t) { return MIS_ERR_UNSUPPORTED; }
