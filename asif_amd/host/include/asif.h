// asif.h -- ASIF::ASIF (include/asif.h:8-101) under the reference's file name; declared in asif_filter.h.
#pragma once
#include "asif_utils.h"
#include "asif_filter.h"
