// asif++.h -- umbrella header, same name as the reference's (include/asif++.h:4-10).
#pragma once
#include "qpwrappers.h"
#include "asif_filter.h"
