// asif_learning_utils.h -- LearningData and the two-network residual (include/asif_learning_utils.h:8-155) under the reference's file name; declared in asif_learning.h.
#pragma once
#include "asif_utils.h"
#include "asif_learning.h"
