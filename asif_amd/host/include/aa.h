// aa.h -- the name the reference's headers and examples include libaffa under (include/asif_robust.h:5,
// include/CyberTimer.hpp:8, lib/libaffa/src/aa.h): here it is the fixed-capacity host restatement of the
// part of that library the filters' model callbacks use (`AAF`, `interval`), asif_affine.h.
#pragma once
#include "asif_affine.h"
