// roar_search_lds_l2.hip -- SearchRoarGraph kernels, visited mode LDS tag table, metric L2
// (one translation unit per pair so that the build runs in parallel; the code is roar_search_v2.h).
#define ROAR_TU_MODE 1
#define ROAR_TU_IP 0
#include "roar_search_v2.h"
