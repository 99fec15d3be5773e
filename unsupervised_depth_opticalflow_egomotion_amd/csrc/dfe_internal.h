// Internal (non-ABI) declarations shared by the translation units of libdfe_hip.so.
#pragma once
#include "../../include/dfe_hip.h"

namespace dfe {
struct ScaleList { float v[DFE_MAX_SCALES]; };
struct IntList { int v[DFE_MAX_SCALES]; };
}  // namespace dfe
