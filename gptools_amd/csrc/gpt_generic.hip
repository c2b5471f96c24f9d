#include "gpt_common.h"
#include "gpt_host.h"
bool gpt_generic_supported(int, int) { return false; }
size_t gpt_generic_workspace_bytes(int, int) { return 0; }
int gpt_generic_col_blocks(int, int) { return 1; }
hipError_t gpt_launch_generic(int, const LatArgs&, cplx*, double*, hipStream_t) { return hipErrorNotSupported; }
