// asif++.h -- umbrella header, same name as the reference's (include/asif++.h:4-10).
// ASIF::ASIF, ASIF::ASIFimplicit, ASIF::ASIFimplicitTB, ASIF::ASIFimplicitRB, ASIF::ASIFrobust,
// ASIF::ASIFrealizable and the solver plug-in ASIF::QPWrapperHip.
#pragma once
#include "qpwrappers.h"
#include "asif_filter.h"
#include "asif_backup_filters.h"
#include "asif_implicit_robust_filter.h"
#include "asif_robust_filter.h"
#include "asif_realizable_filter.h"
