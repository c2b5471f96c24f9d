#include "gpt_common.h"
#include "gpt_host.h"
bool gpt_pow2_supported(int, int) { return false; }
size_t gpt_pow2_workspace_bytes(int, int) { return 0; }
int gpt_pow2_col_blocks(int, int) { return 1; }
hipError_t gpt_launch_pow2(int, const LatArgs&, cplx*, double*, hipStream_t) { return hipErrorNotSupported; }
