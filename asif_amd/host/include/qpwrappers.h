// qpwrappers.h -- solver selection.  Keeps the reference's enumerator name (include/qpwrappers.h:6-9) so
// existing `QPSOLVER::OSQP` call sites and default arguments compile; in this build both names select
// the in-kernel ADMM on the GPU (there is no OSQP and no CPU solver here).
#pragma once
#include <cstdint>
#include "qpwrapper_hip.h"

enum class QPSOLVER : uint8_t { OSQP = 0, HIP = 0 };
