// asif_implicit_robust.h -- ASIF::ASIFimplicitRB (include/asif_implicit_robust.h:19-279) under the reference's file name; declared in asif_implicit_robust_filter.h.
#pragma once
#include "asif_utils.h"
#include "asif_implicit_robust_filter.h"
