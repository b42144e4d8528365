// Library-level entry points of libflowcon_hip.so.
#include <hip/hip_runtime.h>
#include "../../include/flowcon_hip.h"

extern "C" int fc_abi_version(void) { return FC_ABI_VERSION; }
