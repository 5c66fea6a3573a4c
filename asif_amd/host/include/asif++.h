// asif++.h -- umbrella header, same name and same include order as the reference's (include/asif++.h:4-10).
// ASIF::ASIF, ASIF::ASIFrobust, ASIF::ASIFrealizable, ASIF::ASIFimplicit, ASIF::ASIFimplicitRB, ASIF::ASIFimplicitTB,
// the helpers of asif_utils.h and the solver plug-ins (ASIF::QPWrapperHip) -- everything a program written against
// the reference's include/ directory names, so that it compiles with this directory put in its place.
#pragma once
#include "asif_utils.h"
#include "asif.h"
#include "asif_robust.h"
#include "asif_realizable.h"
#include "asif_implicit.h"
#include "asif_implicit_robust.h"
#include "asif_implicit_tb.h"
