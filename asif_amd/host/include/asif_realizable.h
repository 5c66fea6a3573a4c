// asif_realizable.h -- ASIF::ASIFrealizable (include/asif_realizable.h:9-123) under the reference's file name; declared in asif_realizable_filter.h.
#pragma once
#include "asif_utils.h"
#include "asif_realizable_filter.h"
