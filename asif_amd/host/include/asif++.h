// asif++.h -- umbrella header, same name as the reference's (include/asif++.h:4-10).
// ASIF::ASIF, ASIF::ASIFimplicit, ASIF::ASIFimplicitTB and the solver plug-in ASIF::QPWrapperHip.
// ASIFrobust has no C++ mirror yet (its callbacks take libaffa's AAF type); the robust filter is
// reachable through the C ABI (asif_hip_create(..., ASIF_HIP_ROBUST, ...)).
#pragma once
#include "qpwrappers.h"
#include "asif_filter.h"
#include "asif_backup_filters.h"
