// qpwrapper_abstract.h -- ASIF::QPWrapperAbstract under the reference's file name
// (include/qpwrapper_abstract.h:16-51); the class itself is in asif_qp_interface.h.
#pragma once
#include "asif_qp_interface.h"
