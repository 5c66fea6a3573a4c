// qpwrappers.h -- solver selection.  Keeps the reference's enumerator name (include/qpwrappers.h:6-9) so existing
// `QPSOLVER::OSQP` call sites and default arguments compile: in this build OSQP and HIP both select the GPU kernels
// behind ASIF::QPWrapperHip (there is no OSQP here).  HOST, chosen by name only, selects ASIF::QPWrapperHost for the
// single-agent filter() of the classes whose problem has at most three variables and a diagonal cost (class ASIF,
// ASIFimplicit, ASIFimplicitRB, ASIFimplicitTB) -- the product's dual active-set method on the calling thread, no
// launch; a class whose problem is larger (ASIFrobust, ASIFrealizable) keeps QPWrapperHip under either name.
#pragma once
#include <cstdint>
#include "qpwrapper_hip.h"
#include "qpwrapper_host.h"

enum class QPSOLVER : uint8_t { OSQP = 0, HIP = 0, HOST = 1 };

namespace ASIF {
// the solver object a filter class constructs for its (nv, nc, diagonalCost) problem (src/asif.cpp:36-48)
QPWrapperAbstract *makeQPWrapper(const QPSOLVER type, const uint32_t nv, const uint32_t nc, const bool diagonalCost);
} // namespace ASIF
