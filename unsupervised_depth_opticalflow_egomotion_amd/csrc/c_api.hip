// ABI bookkeeping entry points of libdfe_hip.so.
#include "dfe_internal.h"

extern "C" {

int dfe_abi_version(void) { return DFE_ABI_VERSION; }

const char* dfe_error_string(int code) {
  switch (code) {
    case DFE_OK: return "ok";
    case DFE_ERR_NULL: return "a required pointer is NULL";
    case DFE_ERR_DIMS: return "a dimension is out of range";
    case DFE_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    case DFE_ERR_UNSUPPORTED: return "unsupported argument value";
    case DFE_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown dfe error";
  }
}

}  // extern "C"
