// asif_robust.h -- ASIF::ASIFrobust (include/asif_robust.h:11-89) under the reference's file name; declared in asif_robust_filter.h.
#pragma once
#include "asif_utils.h"
#include "asif_robust_filter.h"
