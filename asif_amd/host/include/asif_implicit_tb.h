// asif_implicit_tb.h -- ASIF::ASIFimplicitTB under the reference's file name (include/asif_implicit_tb.h); declared in
// asif_backup_filters.h.  The reference's header ends its include block with `using namespace std;` at global
// scope (include/asif_implicit_tb.h:12-13) and its callers came to rely on it (include/KernelData_70-135kg.h:4 names `vector`
// unqualified, examples call `cout`): kept, in this file only, for sources written against that header.
#pragma once
#include "asif_utils.h"
#include "asif_backup_filters.h"
using namespace std;
