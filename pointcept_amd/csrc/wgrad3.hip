// wgrad3.hip -- the weight gradient of the deep stages' Linear layers (wgrad3.h) as its own translation unit.
#include "ptc_common.h"
#include <stdlib.h>

#define PTC_WGRAD3_IMPL
#include "wgrad3.h"
