// gemm3.hip -- the dense row-wise GEMM of the deep stages (gemm3.h) as its own translation unit (its own compile time and flags).
#include "ptc_common.h"
#include <stdlib.h>

#define PTC_GEMM3_IMPL
#include "gemm3.h"
